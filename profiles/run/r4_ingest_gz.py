# HISTORICAL (kept as the record of how a committed figure was measured): this script sets SKDER_AMD_GPU_INFLATE, a switch the library
# stopped reading in round 5 (include/skder_amd.h lists the live ones) -- on today's tree it would measure the default build under a
# variant's label.  To repeat the measurement check out the round it belongs to (r3_* : round 3, r4_* : round 4).
"""round 4: the gzip ingest with a share of the streams inflated on the device (SKDER_AMD_GPU_INFLATE = percent of the gzip text):
the drop-in on N .fasta.gz sample files, shares 0 / 40 / 60 / 80 / 100, each in its own process, SKDER_AMD_DEBUG phase lines kept"""
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
gz, gzb = bench.gzip_sample_files(paths)
open(os.path.join(tmp, "gz.lst"), "w").write("\n".join(gz))
open(os.path.join(tmp, "plain.lst"), "w").write("\n".join(paths))
res = {}
for kind, share in [("plain", None)] + [("gz", s) for s in os.environ.get("SHARES", "0 40 60 80 100").split()]:
    env = dict(os.environ, SKDER_AMD_DEBUG="1")
    if share is not None:
        env["SKDER_AMD_GPU_INFLATE"] = str(share)
    code = ("import sys,os; sys.path.insert(0,os.getcwd()); import bench\n"
            "ps=open(%r).read().split()\n"
            "bench.end_to_end_sample(%r, ps[:8], 1, 0)\n"
            "bench.end_to_end_sample(%r, ps, 1, 0)\n"
            "b=[bench.end_to_end_sample(%r, ps, 1, 0) for _ in range(3)]\n"
            "print('RESULT', min(x['seconds'] for x in b), b[0]['rows'])\n") % (os.path.join(tmp, kind + ".lst"), tmp, tmp, tmp)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    ing = [l for l in p.stderr.splitlines() if "ingest of %d files" % n in l or "not confirmed" in l]
    sec = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    key = kind if share is None else "gz_device_share_%s" % share
    if sec:
        t = float(sec[-1].split()[1])
        res[key] = {"seconds": t, "text_GB_per_s": nbytes / t / 1e9, "rows": int(sec[-1].split()[2]), "ingest": ing[-2:]}
    else:
        res[key] = {"error": p.stderr[-800:]}
    print(key, res[key], flush=True)
print(json.dumps({"files": n, "fasta_bytes": nbytes, "gz_bytes": gzb, "runs": res}))
