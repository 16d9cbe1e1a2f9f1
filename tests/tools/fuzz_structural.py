"""one-off randomized parity run: GPU engine vs oracle on structurally mutated genomes (not part of the test suite)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle_py as oracle
from skder_amd import engine
import test_gpu_parity as T

REAL = os.environ.get('FUZZ_REAL') == '1'
alpha = np.frombuffer(b"ACGT", np.uint8)
comp = np.zeros(256, np.uint8)
comp[ord("A")], comp[ord("C")], comp[ord("G")], comp[ord("T")] = ord("T"), ord("G"), ord("C"), ord("A")

def derive(rng, anc):
    seq = anc.copy()
    # substitutions
    sub = 10 ** rng.uniform(-3.3, -1.4)
    k = rng.binomial(len(seq), sub)
    if k:
        idx = rng.choice(len(seq), k, replace=False)
        seq[idx] = alpha[(np.searchsorted(alpha, seq[idx]) + 1 + rng.randint(0, 3, k)) % 4]
    # indels
    every = int(10 ** rng.uniform(2.3, 3.9)) if not REAL else max(150, int(12.0 / sub))   # one indel per ~12 substitutions
    out, pos = [], 0
    while pos < len(seq):
        step = rng.randint(every // 2, every * 2)
        out.append(seq[pos:pos + step]); pos += step
        n_ = rng.geometric(0.4) if REAL else rng.randint(1, 60)
        if rng.rand() < 0.5: out.append(alpha[rng.randint(0, 4, n_)])
        else: pos += n_
    seq = np.concatenate(out)
    # structural events
    for _ in range(rng.randint(0, 12) if not REAL else rng.randint(0, 4)):
        L = len(seq); a = rng.randint(0, L - 25000); n = rng.randint(1000, 20000); ev = rng.randint(0, 4)
        seg = seq[a:a + n]
        if ev == 0: seq = np.concatenate([seq[:a], comp[seg[::-1]], seq[a + n:]])                 # inversion
        elif ev == 1:                                                                            # translocation
            rest = np.concatenate([seq[:a], seq[a + n:]]); b = rng.randint(0, len(rest)); seq = np.concatenate([rest[:b], seg, rest[b:]])
        elif ev == 2:                                                                            # tandem / dispersed duplication
            m = rng.randint(1000, 5000); d = seq[a:a + m]
            for _ in range(rng.randint(1, 6)):
                b = rng.randint(0, len(seq)); seq = np.concatenate([seq[:b], d, seq[b:]])
        else: seq = np.concatenate([seq[:a], seq[a + n:]])                                       # deletion
    # records
    nrec = int(10 ** rng.uniform(0, 1.9))
    cuts = np.sort(rng.choice(np.arange(600, len(seq) - 600), size=min(nrec - 1, 80), replace=False)) if nrec > 1 else np.array([], int)
    bounds = np.concatenate([[0], cuts, [len(seq)]])
    lens = np.diff(bounds)
    keep = [lens[0]]
    for l in lens[1:]:
        if l < 500 or keep[-1] < 500: keep[-1] += l
        else: keep.append(l)
    return seq, np.array(keep, np.uint32)

def main():
    ctx = engine.Context(0)
    gpu = (engine, ctx, torch)
    p = oracle.default_params()
    bad = 0
    t0 = time.time()
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        rng = np.random.RandomState(seed)
        anc = alpha[rng.randint(0, 4, rng.randint(300000, 900000))]
        gl = [derive(rng, anc) for _ in range(5)]
        bases, lens = [g[0] for g in gl], [g[1] for g in gl]
        s, _ = T._sketch(gpu, lens, bases)
        og = [oracle.Genome.from_bases(b, l, p) for b, l in zip(bases, lens)]
        edges = s.triangle_rows(0, 1, 0.0)
        want = T._oracle_edges(oracle, og, p, 0.0)
        c = ctx.counters()
        try:
            T._check_edges(edges, want)
            print("seed", seed, "ok", len(want), "pairs; chunks", int(c[0]), "slow", int(c[1]), flush=True)
        except AssertionError as e:
            bad += 1
            print("seed", seed, "MISMATCH", str(e)[:300], flush=True)
        s.close()
    print("done", bad, "mismatches in", round(time.time() - t0, 1), "s")

if __name__ == '__main__':
    main()
