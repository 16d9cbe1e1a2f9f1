# HISTORICAL (kept as the record of how a committed figure was measured): this script sets SKDER_AMD_ZLIB, a switch the library
# stopped reading in round 5 (include/skder_amd.h lists the live ones) -- on today's tree it would measure the default build under a
# variant's label.  To repeat the measurement check out the round it belongs to (r3_* : round 3, r4_* : round 4).
"""ingest with the FASTA parse on the device against the host parse: the drop-in on 256 and 1024 sample files (plain, then gzip),
two-point (fixed + marginal) as bench.py's end_to_end does; each leg in its own process, SKDER_AMD_DEBUG phase lines kept"""
import json, os, subprocess, sys
sys.path.insert(0, os.getcwd())
import bench, torch
from skder_amd import engine, synth
ctx = engine.Context(0)
n = int(os.environ.get("N", "1024"))
recipe = synth.make_recipe(n, genome_len=3_000_000)
layout = engine.BatchLayout(recipe.rec_lens)
d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
ctx.synth_fill(d.data_ptr(), layout, recipe.lineage, recipe.params)
tmp, paths, nbytes = bench.write_sample_files([(layout, d)], n)
del d
gz, _ = bench.gzip_sample_files(paths)
res = {}
for kind, ps in (("plain", paths), ("gz", gz)):
    open(os.path.join(tmp, kind + ".lst"), "w").write("\n".join(ps))
    for mode, extra in (("device_parse", {}), ("host_parse", {"SKDER_AMD_HOST_PARSE": "1"})) + ((("device_parse_zlib", {"SKDER_AMD_ZLIB": "1"}),) if kind == "gz" else ()):
        env = dict(os.environ, SKDER_AMD_DEBUG="1", **extra)
        code = ("import sys,os; sys.path.insert(0,os.getcwd()); import bench\n"
                "ps=open(%r).read().split()\n"
                "bench.end_to_end_sample(%r, ps[:8], 1, 0)\n"
                "bench.end_to_end_sample(%r, ps, 1, 0)\n"
                "a=min(bench.end_to_end_sample(%r, ps[:len(ps)//4], 1, 0)['seconds'] for _ in range(3))\n"
                "b=min(bench.end_to_end_sample(%r, ps, 1, 0)['seconds'] for _ in range(3))\n"
                "print('RESULT', a, b)\n") % (os.path.join(tmp, kind + ".lst"), tmp, tmp, tmp, tmp)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        ing = [l for l in p.stderr.splitlines() if "ingest of %d files" % n in l]
        sec = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if sec:
            a, b = map(float, sec[-1].split()[1:])
            marg = (nbytes * 0.75) / (b - a)
            res[kind + "_" + mode] = {"s_quarter": a, "s_all": b, "marginal_MB_per_s": marg / 1e6, "fixed_s": a - nbytes * 0.25 / marg, "ingest": ing[-3:]}
        else:
            res[kind + "_" + mode] = {"error": p.stderr[-500:]}
        print(kind, mode, res[kind + "_" + mode], flush=True)
print(json.dumps({"files": n, "fasta_bytes": nbytes, "runs": res}))
