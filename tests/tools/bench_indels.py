"""throughput on genomes WITH indels and structural variants (host-generated; the counter-based generator has none)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from skder_amd import engine
import test_gpu_parity as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
alpha = np.frombuffer(b"ACGT", np.uint8)
rng = np.random.RandomState(7)
anc = alpha[rng.randint(0, 4, L)]
t0 = time.time()
fam = [T._structural_variant(rng, anc, True) for _ in range(N)]
print("generated", N, "genomes in", round(time.time() - t0, 1), "s", flush=True)
ctx = engine.Context(0)
bases, lens = [g[0] for g in fam], [g[1] for g in fam]
s, _ = T._sketch((engine, ctx, torch), lens, bases)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    edges = s.triangle_rows(0, 1, 80.0, copy=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t = ctx.timing(); c = ctx.counters()
    print("triangle %.1f ms, %d edges, chunks %d slow %d (%.1f %%); join %.2f fast %.2f slow %.2f fin %.2f ms" % (dt * 1e3, len(edges), c[0], c[1], 100.0 * c[1] / max(c[0], 1), c[2] / 1000.0, t[3], t[4], t[5]), flush=True)
