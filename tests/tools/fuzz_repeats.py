"""randomized parity, wider: sizes, divergence, repeats, non-ACGT, triangle + rectangle"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle_py as oracle
from skder_amd import engine
import test_gpu_parity as T
alpha = np.frombuffer(b"ACGT", np.uint8)
comp = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTacgtN", b"TGCAtgcaN"): comp[a] = b

MODE = sys.argv[3] if len(sys.argv) > 3 else ''
def ancestor(rng):
    L = int(10 ** rng.uniform(4.7, 6.2)) if MODE != 'big' else int(rng.uniform(2.0e6, 5.2e6))
    if MODE == 'tiny': L = int(10 ** rng.uniform(3.1, 4.7))
    seq = alpha[rng.randint(0, 4, L)]
    # repeats: a few segment families copied around, tandem repeats, low-complexity stretches
    for _ in range(rng.randint(0, 6)):
        if L < 13000: break
        m = rng.randint(300, 6000); a = rng.randint(0, L - m); d = seq[a:a + m].copy()
        for _ in range(rng.randint(1, 10)):
            b = rng.randint(0, L - m); seq[b:b + m] = d if rng.rand() < 0.7 else comp[d[::-1]]
    if MODE == 'rep':
        m = rng.randint(800, 5000); d = seq[:m].copy()
        for _ in range(rng.randint(40, 300)):
            b = rng.randint(0, L - m); seq[b:b + m] = d if rng.rand() < 0.8 else comp[d[::-1]]
    for _ in range(rng.randint(0, 4)):
        unit = alpha[rng.randint(0, 4, rng.randint(1, 40))]; reps = rng.randint(5, 400)
        t = np.tile(unit, reps)[:min(20000, L // 2)]; b = rng.randint(0, L - len(t)); seq[b:b + len(t)] = t
    return seq

def descend(rng, anc, level):
    seq = anc.copy()
    sub = 10 ** rng.uniform(-4.0, -0.9) if level else 0.0
    k = rng.binomial(len(seq), sub)
    if k:
        idx = rng.choice(len(seq), k, replace=False)
        seq[idx] = alpha[(np.searchsorted(alpha, seq[idx]) + 1 + rng.randint(0, 3, k)) % 4]
    if level:
        every = max(100, int(rng.uniform(5, 30) / max(sub, 1e-5))) if rng.rand() < 0.7 else int(10 ** rng.uniform(2.2, 3.5))
        geo = rng.rand() < 0.7
        out, pos = [], 0
        while pos < len(seq):
            step = rng.randint(every // 2 + 1, every * 2 + 2); out.append(seq[pos:pos + step]); pos += step
            n = rng.geometric(0.4) if geo else rng.randint(1, 80)
            if rng.rand() < 0.5: out.append(alpha[rng.randint(0, 4, n)])
            else: pos += n
        seq = np.concatenate(out)
        for _ in range(rng.randint(0, 8)):
            if len(seq) < 60000: break
            a, n, ev = rng.randint(0, len(seq) - 25000), rng.randint(500, 20000), rng.randint(0, 4)
            seg = seq[a:a + n]
            if ev == 0: seq = np.concatenate([seq[:a], comp[seg[::-1]], seq[a + n:]])
            elif ev == 1:
                rest = np.concatenate([seq[:a], seq[a + n:]]); b = rng.randint(0, len(rest)); seq = np.concatenate([rest[:b], seg, rest[b:]])
            elif ev == 2:
                d = seq[a:a + rng.randint(500, 5000)]
                for _ in range(rng.randint(1, 6)):
                    b = rng.randint(0, len(seq)); seq = np.concatenate([seq[:b], d, seq[b:]])
            else: seq = np.concatenate([seq[:a], seq[a + n:]])
    # non-ACGT: N runs and lower case
    seq = seq.copy()
    for _ in range(rng.randint(0, 5)):
        a = rng.randint(0, max(1, len(seq) - 100)); n = rng.randint(1, 3000); seq[a:a + n] = ord("N")
    if rng.rand() < 0.3:
        a = rng.randint(0, max(1, len(seq) - 100)); n = rng.randint(1, max(2, len(seq) // 3)); seq[a:a + n] |= 0x20
    nrec = int(10 ** rng.uniform(0, 2.2))
    lo = 600
    cuts = np.sort(rng.choice(np.arange(lo, len(seq) - lo), size=min(nrec - 1, 150, len(seq) - 2 * lo), replace=False)) if nrec > 1 and len(seq) > 2 * lo + 10 else np.array([], int)
    lens = np.diff(np.concatenate([[0], cuts, [len(seq)]]))
    keep = [lens[0]]
    for l in lens[1:]:
        if l < 500 or keep[-1] < 500: keep[-1] += l
        else: keep.append(l)
    return seq, np.array(keep, np.uint32)

def main():
    ctx = engine.Context(0); gpu = (engine, ctx, torch); p = oracle.default_params()
    bad = 0; t0 = time.time()
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        rng = np.random.RandomState(seed)
        anc = ancestor(rng)
        n = rng.randint(3, 7) if MODE != 'big' else 3
        if MODE == 'batch': os.environ['SKDER_AMD_CHUNK_BUDGET'] = str(rng.randint(50, 3000))
        gl = [descend(rng, anc, i) for i in range(n)]
        bases, lens = [g[0] for g in gl], [g[1] for g in gl]
        try:
            if MODE == 'append':      # several sketch calls (random split into batches) instead of one
                s = engine.Sketches(ctx); pos = 0
                while pos < n:
                    k = rng.randint(1, n - pos + 1)
                    lay = engine.BatchLayout(lens[pos:pos + k]); d = torch.from_numpy(lay.pack_host(bases[pos:pos + k])).cuda()
                    s.sketch_batch(d.data_ptr(), lay); torch.cuda.synchronize(); pos += k
            else:
                s, _ = T._sketch(gpu, lens, bases)
            og = [oracle.Genome.from_bases(b, l, p) for b, l in zip(bases, lens)]
            T._compare_sketch(engine, s, oracle, og)
            screen = 0.0 if rng.rand() < 0.7 else 80.0
            edges = s.triangle_rows(0, 1, screen)
            want = T._oracle_edges(oracle, og, p, screen)
            T._check_edges(edges, want)
            c = ctx.counters()
            # rectangle: the last two genomes as queries against all
            q, _ = T._sketch(gpu, lens[-2:], bases[-2:])
            rect = s.rectangle(q, screen)
            got = {(int(e["ref"]), int(e["query"])): e for e in rect}
            for r in range(n):
                for qi in range(2):
                    g = n - 2 + qi
                    ok, _ = oracle.screen(og[r], og[g], screen, p)
                    pr = oracle.pair(og[r], og[g], p) if ok else None
                    if pr is not None and pr.n_chains and pr.ani > 0:
                        e = got[(r, qi)]
                        assert int(e["cell_seeds"]) == pr.cell_seeds and float(e["ani"]) == pr.ani and int(e["sum_seeds"]) == pr.sum_seeds, ("rect", r, qi)
                        assert float(e["af_ref"]) == pr.af_ref and float(e["af_query"]) == pr.af_query, ("rect af", r, qi)
                    else:
                        assert (r, qi) not in got, ("rect extra", r, qi)
            q.close(); s.close()
            print("seed", seed, "ok", len(want), "pairs, L", len(anc), "chunks", int(c[0]), "slow", int(c[1]), flush=True)
        except AssertionError as e:
            bad += 1; print("seed", seed, "MISMATCH", str(e)[:300], flush=True)
    print("done", bad, "mismatches in", round(time.time() - t0, 1), "s")
if __name__ == "__main__":
    main()
