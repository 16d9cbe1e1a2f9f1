"""Host-side mirror of the skani call sites of /root/reference/src/skDER/skder.py.

Same function names, argument order and failure behaviour as the reference (RuntimeError when the
expected output is not produced, cf. util.runCmd, /root/reference/src/skDER/util.py:636-652), but
instead of `subprocess.call('skani ...')` each one calls the C ABI of libskder_amd.so."""
import ctypes as C
import os

from . import _lib

# skani's own defaults for `dist` / `search` when the caller passes no flag
SKANI_DEFAULT_SCREEN = 80.0
SKANI_DEFAULT_MIN_AF = 15.0


def _device() -> int:
    return int(os.environ.get("SKDER_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def _devices():
    """GPUs of the drop-in entry points: SKDER_AMD_DEVICES="0,1,2,3" spreads one call over several (skani's `-t T` asks
    for the whole machine, skder.py:18); default: the single GPU of _device()"""
    v = os.environ.get("SKDER_AMD_DEVICES", "").strip()
    return [int(x) for x in v.split(",") if x.strip() != ""] if v else [_device()]


def _c_devices(devices):
    return (C.c_int * len(devices))(*devices), len(devices)


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
                     aligned_fraction_cutoff, selection_mode, test_cutoffs_flag, logObject, threads=1, n50_file=None):
    """Replaces skder.py:10-28: `skani triangle -l LIST --min-af AF -E <params> -t T -o OUT`.
    n50_file (optional): also write Concatenated_N50.txt from the same ingest pass."""
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
        devs, nd = _c_devices(_devices())
        rc = _lib.lib().skder_amd_triangle_multi(genome_listing_file.encode(), min_af, screen, devs, nd,
                                                 skani_result_file.encode(), n50_file.encode() if n50_file else None,
                                                 err, _lib.ERRLEN)
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
        rep_listing_file = cluster_dir + 'Reps_Listing.txt'
        nonrep_listing_file = cluster_dir + 'NonReps_Listing.txt'
        _split_listing(genome_listing_file, skder_result_file, rep_listing_file, nonrep_listing_file)
        screen = parse_skani_params(skani_dist_parameters)
        err = C.create_string_buffer(_lib.ERRLEN)
        rc = _lib.lib().skder_amd_dist(rep_listing_file.encode(), nonrep_listing_file.encode(), SKANI_DEFAULT_MIN_AF,
                                       screen, _device(), skani_result_file.encode(), err, _lib.ERRLEN)
        if rc != 0 or not os.path.isfile(skani_result_file):
            raise RuntimeError('Had an issue running: skder_amd_dist: %s' % err.value.decode())
    except Exception as e:
        raise RuntimeError('Error running skani dist command: %s' % e)


def _split_listing(listing_file, reps_file, reps_out, others_out):
    """Reps_Listing.txt / NonReps_Listing.txt of skder.py:36-56: a genome of the listing is a representative when the
    last component of its path equals the last component of a line of the result file (the reference compares base
    names, so two genomes with one base name in different directories count as one)."""
    rep_names = {os.path.basename(l.rstrip('\n').strip()) for l in open(reps_file) if l.strip()}
    with open(listing_file) as src, open(reps_out, 'w') as reps, open(others_out, 'w') as others:
        for path in (l.strip() for l in src):
            (reps if path.rsplit('/', 1)[-1] in rep_names else others).write(path + '\n')


def _read_n50_table(path):
    """Concatenated_N50.txt -> [(genome path, N50 as float)] in file order (skder.py:106-111 parses the N50 with float())"""
    table = []
    for row in open(path):
        genome, _, n50 = row.rstrip('\n').strip().partition('\t')
        table.append((genome, float(n50)))
    return table


def _accounted_by_search_table(tsv, ani_cutoff, af_cutoff):
    """the Ref_file entries of a `skani search` table that skder.py:122-129 marks as accounted for: third column >= the
    ANI cut-off and FIFTH column >= the AF cut-off (the reference names that column align_frac_ref; in the table's header
    it is Align_fraction_query)"""
    hit = set()
    with open(tsv) as f:
        next(f, None)                                   # header
        for row in f:
            col = row.rstrip('\n').split('\t')
            if len(col) == 7 and float(col[2]) >= ani_cutoff and float(col[4]) >= af_cutoff:
                hit.add(col[0])
    return hit


class Database:
    """The sketch database resident in HBM (section C of include/skder_amd.h): built from a listing
    (N50s computed in the same pass, util.py:686-724) or loaded from a sketch store."""

    def __init__(self, handle):
        self._h = handle
        L = _lib.lib()
        n = L.skder_amd_db_size(handle)
        self.paths = [L.skder_amd_db_path(handle, i).decode() for i in range(n)]
        self.n50 = [int(L.skder_amd_db_n50(handle, i)) for i in range(n)]

    @classmethod
    def from_listing(cls, listing_file, n50_file=None, device=None, devices=None):
        """devices: several GPU indices -> the database is spread over them (include/skder_amd.h, skder_amd_sketch_multi)"""
        err = C.create_string_buffer(_lib.ERRLEN)
        if devices is None:
            devices = _devices() if device is None else [device]
        devs, nd = _c_devices(list(devices))
        h = _lib.lib().skder_amd_sketch_multi(listing_file.encode(), devs, nd, n50_file.encode() if n50_file else None, err, _lib.ERRLEN)
        if not h:
            raise RuntimeError('Had an issue running: skder_amd_sketch: %s' % err.value.decode())
        return cls(h)

    @classmethod
    def from_sketches(cls, sketches, paths, n50=None, first_names=None, device=None):
        """a database from a sketch set already resident in HBM (skder_amd.engine.Sketches): the raw sketches are copied into
        a database of their own; `sketches` stays the caller's (include/skder_amd.h, skder_amd_db_from_sketches)"""
        import numpy as np
        k = len(paths)
        if (first_names is not None and len(first_names) != k) or (n50 is not None and len(n50) != k):
            raise ValueError("paths, first_names and n50 must have one entry per genome")
        ps = (C.c_char_p * max(k, 1))(*[p.encode() for p in paths])
        fn = (C.c_char_p * max(k, 1))(*[f.encode() for f in first_names]) if first_names is not None else None
        n50a = np.ascontiguousarray(n50, np.uint64) if n50 is not None else None
        err = C.create_string_buffer(_lib.ERRLEN)
        h = _lib.lib().skder_amd_db_from_sketches(sketches.h, _device() if device is None else device, k, ps, fn,
                                                  n50a.ctypes.data if n50a is not None else None, err, _lib.ERRLEN)
        if not h:
            raise RuntimeError('Had an issue running: skder_amd_db_from_sketches: %s' % err.value.decode())
        return cls(h)

    @classmethod
    def load(cls, store_file, device=None):
        err = C.create_string_buffer(_lib.ERRLEN)
        h = _lib.lib().skder_amd_db_load(store_file.encode(), _device() if device is None else device, err, _lib.ERRLEN)
        if not h:
            raise RuntimeError('Had an issue running: skder_amd_db_load: %s' % err.value.decode())
        return cls(h)

    def save(self, store_file):
        err = C.create_string_buffer(_lib.ERRLEN)
        if _lib.lib().skder_amd_db_save(self._h, store_file.encode(), err, _lib.ERRLEN) != 0 or not os.path.isfile(store_file):
            raise RuntimeError('Had an issue running: skder_amd_db_save: %s' % err.value.decode())

    @staticmethod
    def _rows(p, n, copy=True):
        """copy=False: a view of the library's row buffer, valid until the next call on this database"""
        import numpy as np
        from .engine import EDGE_DTYPE
        if n.value == 0:
            return np.zeros(0, EDGE_DTYPE)
        v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value * EDGE_DTYPE.itemsize,)).view(EDGE_DTYPE)
        return v.copy() if copy else v

    def triangle(self, min_af, screen, out_tsv=None):
        """rows of `skani triangle` over the database, in skani's order, as an edge array"""
        p, n = C.POINTER(_lib.Edge)(), C.c_uint64()
        err = C.create_string_buffer(_lib.ERRLEN)
        rc = _lib.lib().skder_amd_db_triangle(self._h, float(min_af), float(screen), out_tsv.encode() if out_tsv else None,
                                              C.byref(p), C.byref(n), err, _lib.ERRLEN)
        if rc != 0 or (out_tsv and not os.path.isfile(out_tsv)):
            raise RuntimeError('Had an issue running: skder_amd_db_triangle: %s' % err.value.decode())
        return self._rows(p, n)

    def search_batch(self, queries, min_af=SKANI_DEFAULT_MIN_AF, screen=SKANI_DEFAULT_SCREEN, out_tsvs=None, live=None, copy=True):
        """rows of `skani search` for several queries at once; `query` = position in `queries`.  live: one flag per database genome
        (uint8 / bool array), 0 = the caller has no use for rows of that genome (they are not computed).  copy=False: the rows are a
        view of the library's buffer, valid until the next call on this database (a one-species search returns gigabytes of rows)"""
        k = len(queries)
        qs = (C.c_char_p * max(k, 1))(*[q.encode() for q in queries])
        outs = None
        if out_tsvs is not None:
            outs = (C.c_char_p * max(k, 1))(*[o.encode() if o else None for o in out_tsvs])
        p, n = C.POINTER(_lib.Edge)(), C.c_uint64()
        err = C.create_string_buffer(_lib.ERRLEN)
        mask = None
        if live is not None:
            import numpy as np
            mask = np.ascontiguousarray(live, np.uint8)
            if len(mask) != len(self.paths):
                raise ValueError("live: one flag per database genome")
        rc = _lib.lib().skder_amd_search_batch_live(self._h, qs, k, float(min_af), float(screen), outs,
                                                    mask.ctypes.data if mask is not None else None, C.byref(p), C.byref(n), err, _lib.ERRLEN)
        if rc != 0:
            raise RuntimeError('Had an issue running: skder_amd_search_batch: %s' % err.value.decode())
        return self._rows(p, n, copy)

    def close(self):
        if self._h:
            try:
                _lib.lib().skder_amd_db_free(self._h)
            except Exception:
                pass
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        self.close()


def text_value(x):
    """the number the reference would parse from skani's table for a fraction x: float('%.2f' % (f32(x)*100))"""
    import numpy as np
    return float('%.2f' % float(np.float32(x) * np.float32(100)))


def lowMemGreedyDerep(all_genomes_listing_file, skder_lm_workspace, concat_n50_result_file, skder_result_file, outdir,
                      ani_cutoff, af_cutoff, logObject, mge_proc_to_unproc_mapping=None, threads=1, search_batch=None,
                      database=None):
    """Replaces skder.py:95-134: `skani sketch` once, then one `skani search` per representative in
    N50-descending order.  The sketch database lives in HBM for the whole loop.

    search_batch=1 reproduces the reference loop call for call (each search writes the 7-column TSV,
    which is parsed by the reference's code).  Otherwise (default) the rows of the
    next few unaccounted candidates are computed speculatively in one pass and applied in order;
    rows of candidates that an earlier member of the batch accounted for are discarded, so the
    result equals the sequential loop's.  The speculative searches also leave out what cannot change the result: the
    reference's loop only ever ADDS the Ref of a row to `accounted_genomes` (skder.py:127-129), so rows of genomes that are
    accounted for already, or that were handled earlier in the order, are never computed (Database.search_batch(live=...));
    SKDER_AMD_SEARCH_ALL=1 computes them anyway (A/B; the listing is the same)."""
    if search_batch is None:
        search_batch = int(os.environ.get('SKDER_AMD_SEARCH_BATCH', '0'))
    db = database if database is not None else Database.from_listing(all_genomes_listing_file)
    try:
        # N50 descending; Python's sort is stable, so equal N50s keep the listing's order (skder.py:116)
        order = sorted(_read_n50_table(concat_n50_result_file), key=lambda gn: gn[1], reverse=True)
        skder_result_handle = open(skder_result_file, 'w')
        accounted_genomes = set([])

        def emit(genome):
            name = genome
            if mge_proc_to_unproc_mapping is not None:
                name = mge_proc_to_unproc_mapping[genome]
            skder_result_handle.write(name + '\n')

        if search_batch == 1:
            err = C.create_string_buffer(_lib.ERRLEN)
            for gn in order:
                if gn[0] in accounted_genomes:
                    continue
                skani_search_result = skder_lm_workspace + 'current_search_results.tsv'
                rc = _lib.lib().skder_amd_search(db._h, gn[0].encode(), SKANI_DEFAULT_MIN_AF, SKANI_DEFAULT_SCREEN,
                                                 skani_search_result.encode(), err, _lib.ERRLEN)
                if rc != 0 or not os.path.isfile(skani_search_result):
                    raise RuntimeError('Had an issue running: skder_amd_search %s: %s' % (gn[0], err.value.decode()))
                accounted_genomes |= _accounted_by_search_table(skani_search_result, ani_cutoff, af_cutoff)
                emit(gn[0])
        else:
            # rows are tested in bulk by the native code (the table's two-decimal text, skder.py:127-129) and applied with numpy:
            # one species with 20,000 genomes returns ~20,000 rows per search, and per-row Python was the loop's time
            import numpy as np
            index_of = {p: i for i, p in enumerate(db.paths)}
            acc = np.zeros(len(db.paths), bool)
            handled = np.zeros(len(db.paths), bool)                # candidates whose turn has come (representative or not)
            live_only = os.environ.get('SKDER_AMD_SEARCH_ALL') != '1'
            extra = set()                                         # (genomes of the N50 table that are not in the database: cannot be accounted for)
            is_acc = lambda g: acc[index_of[g]] if g in index_of else g in extra
            width = search_batch if search_batch > 1 else 4       # 0: adaptive, starting at 4
            pos = 0
            import time as _time
            stats = {"searches": 0, "rows": 0, "batches": 0, "search_s": 0.0, "rows_pass_s": 0.0}
            while pos < len(order):
                batch = []
                while pos < len(order) and len(batch) < width:
                    if not is_acc(order[pos][0]):
                        batch.append(order[pos][0])
                    pos += 1
                if not batch:
                    break
                _t0 = _time.perf_counter()
                rows = db.search_batch(batch, live=~(acc | handled) if live_only else None, copy=False)    # consumed before the next search
                stats["search_s"] += _time.perf_counter() - _t0
                for genome in batch:
                    if genome in index_of:
                        handled[index_of[genome]] = True
                ok = np.zeros(len(rows), np.uint8)
                _t0 = _time.perf_counter()
                if len(rows):
                    _lib.lib().skder_amd_rows_pass(rows.ctypes.data, len(rows), float(ani_cutoff), float(af_cutoff), 5, ok.ctypes.data)
                stats["rows_pass_s"] += _time.perf_counter() - _t0
                bounds = np.searchsorted(rows['query'], np.arange(len(batch) + 1))          # rows come grouped by query, in order
                stats["batches"] += 1
                kept = 0
                for k, genome in enumerate(batch):
                    if not is_acc(genome):
                        lo, hi = int(bounds[k]), int(bounds[k + 1])
                        acc[rows['ref'][lo:hi][ok[lo:hi] != 0]] = True
                        emit(genome)
                        kept += 1
                        stats["searches"] += 1
                        stats["rows"] += hi - lo
                if search_batch == 0:
                    if kept == len(batch):
                        width = min(width * 2, 64)
                    elif kept * 2 < len(batch):
                        width = max(width // 2, 1)
            lowMemGreedyDerep.last_stats = stats
        skder_result_handle.close()
    finally:
        if database is None:
            db.close()
