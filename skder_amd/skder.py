"""Host-side mirror of the skani call sites of /root/reference/src/skDER/skder.py.

Same function names, argument order and failure behaviour as the reference (RuntimeError when the
expected output is not produced, cf. util.runCmd, /root/reference/src/skDER/util.py:636-652), but
instead of `subprocess.call('skani ...')` each one calls the C ABI of libskder_amd.so."""
import ctypes as C
import os
from operator import itemgetter

from . import _lib

# skani's own defaults for `dist` / `search` when the caller passes no flag
SKANI_DEFAULT_SCREEN = 80.0
SKANI_DEFAULT_MIN_AF = 15.0


def _device() -> int:
    return int(os.environ.get("SKDER_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def parse_skani_params(params: str, default_screen: float = SKANI_DEFAULT_SCREEN) -> float:
    """The `-p` string of bin/skder:132,199-201: '-s <float>' only; anything else raises."""
    screen = C.c_double(default_screen)
    err = C.create_string_buffer(_lib.ERRLEN)
    if _lib.lib().skder_amd_parse_skani_params((params or "").encode(), C.byref(screen), err, _lib.ERRLEN) != 0:
        raise RuntimeError(err.value.decode())
    return screen.value


def _log(logObject, level, msg):
    if logObject is not None:
        getattr(logObject, level)(msg)


def runSkaniTriangle(genome_listing_file, skani_result_file, skani_triangle_parameters,
                     aligned_fraction_cutoff, selection_mode, test_cutoffs_flag, logObject, threads=1):
    """Replaces skder.py:10-28: `skani triangle -l LIST --min-af AF -E <params> -t T -o OUT`."""
    try:
        min_af = float(aligned_fraction_cutoff)
        if test_cutoffs_flag:   # skder.py:19-25
            min_af = 10.0
            if selection_mode == 'dynamic':
                min_af = max([min_af - 20.0, 0.0])
        screen = parse_skani_params(skani_triangle_parameters)
        what = 'skder_amd_triangle(%s, min_af=%s, screen=%s) -> %s' % (genome_listing_file, min_af, screen, skani_result_file)
        _log(logObject, 'info', 'Running %s' % what)
        err = C.create_string_buffer(_lib.ERRLEN)
        rc = _lib.lib().skder_amd_triangle(genome_listing_file.encode(), min_af, screen, _device(),
                                           skani_result_file.encode(), err, _lib.ERRLEN)
        if rc != 0 or not os.path.isfile(skani_result_file):
            _log(logObject, 'error', 'Had an issue running: %s: %s' % (what, err.value.decode()))
            raise RuntimeError('Had an issue running: %s: %s' % (what, err.value.decode()))
        _log(logObject, 'info', 'Successfully ran: %s' % what)
    except Exception as e:
        raise RuntimeError('Error running skani triangle command: %s' % e)


def runSkaniDist(cluster_dir, skder_result_file, genome_listing_file, skani_result_file, skani_dist_parameters,
                 aligned_fraction_cutoff, selection_mode, test_cutoffs_flag, logObject, threads=1):
    """Replaces skder.py:30-63: representatives vs non-representatives with `skani dist`.
    As in the reference, aligned_fraction_cutoff is NOT forwarded (skani's default min-af applies)."""
    try:
        rep_file_names = set([])
        with open(skder_result_file) as orf:
            for line in orf:
                line = line.strip()
                rep_file_names.add(line.split('/')[-1])
        rep_listing_file = cluster_dir + 'Reps_Listing.txt'
        nonrep_listing_file = cluster_dir + 'NonReps_Listing.txt'
        with open(rep_listing_file, 'w') as rlf, open(nonrep_listing_file, 'w') as nlf, open(genome_listing_file) as oglf:
            for line in oglf:
                line = line.strip()
                base_name = line.split('/')[-1]
                if base_name in rep_file_names:
                    rlf.write(line + '\n')
                else:
                    nlf.write(line + '\n')
        screen = parse_skani_params(skani_dist_parameters)
        err = C.create_string_buffer(_lib.ERRLEN)
        rc = _lib.lib().skder_amd_dist(rep_listing_file.encode(), nonrep_listing_file.encode(), SKANI_DEFAULT_MIN_AF,
                                       screen, _device(), skani_result_file.encode(), err, _lib.ERRLEN)
        if rc != 0 or not os.path.isfile(skani_result_file):
            raise RuntimeError('Had an issue running: skder_amd_dist: %s' % err.value.decode())
    except Exception as e:
        raise RuntimeError('Error running skani dist command: %s' % e)


def lowMemGreedyDerep(all_genomes_listing_file, skder_lm_workspace, concat_n50_result_file, skder_result_file, outdir,
                      ani_cutoff, af_cutoff, logObject, mge_proc_to_unproc_mapping=None, threads=1):
    """Replaces skder.py:95-134: `skani sketch` once, then one `skani search` per representative in
    N50-descending order.  The sketch database lives in HBM for the whole loop; each search writes
    the same 7-column TSV the reference parses, and the parsing below is the reference's."""
    err = C.create_string_buffer(_lib.ERRLEN)
    db = _lib.lib().skder_amd_sketch(all_genomes_listing_file.encode(), _device(), err, _lib.ERRLEN)
    if not db:
        raise RuntimeError('Had an issue running: skder_amd_sketch: %s' % err.value.decode())
    try:
        n50_data = []
        with open(concat_n50_result_file) as ocnrf:
            for line in ocnrf:
                line = line.strip()
                genome, n50 = line.split('\t')
                n50_data.append([genome, float(n50)])
        skder_result_handle = open(skder_result_file, 'w')
        accounted_genomes = set([])
        for gn in sorted(n50_data, key=itemgetter(1), reverse=True):
            if gn[0] in accounted_genomes:
                continue
            skani_search_result = skder_lm_workspace + 'current_search_results.tsv'
            rc = _lib.lib().skder_amd_search(db, gn[0].encode(), SKANI_DEFAULT_MIN_AF, SKANI_DEFAULT_SCREEN,
                                             skani_search_result.encode(), err, _lib.ERRLEN)
            if rc != 0 or not os.path.isfile(skani_search_result):
                raise RuntimeError('Had an issue running: skder_amd_search %s: %s' % (gn[0], err.value.decode()))
            with open(skani_search_result) as ossr:
                for i, line in enumerate(ossr):
                    if i == 0:
                        continue
                    line = line.strip('\n')
                    ref_file, query_file, ani, align_frac_query, align_frac_ref, ref_name, query_name = line.split('\t')
                    if float(ani) >= ani_cutoff and float(align_frac_ref) >= af_cutoff:
                        accounted_genomes.add(ref_file)
            name = gn[0]
            if mge_proc_to_unproc_mapping is not None:
                name = mge_proc_to_unproc_mapping[gn[0]]
            skder_result_handle.write(name + '\n')
        skder_result_handle.close()
    finally:
        _lib.lib().skder_amd_db_free(db)
