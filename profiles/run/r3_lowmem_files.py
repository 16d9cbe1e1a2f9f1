"""low_mem_greedy (skder.py:95-134: the reference's one published workload, README.md:27) FROM FASTA FILES: N synthetic genomes of
BASELINE config 4's shape written as FASTA files (plain; GZ=1: gzip level 1), then listing -> ingest (read, PCIe, device parse, N50,
sketch, index: Database.from_listing with the N50 table) -> lowMemGreedyDerep -i 99.5 -f 50 with the database resident."""
import gzip, json, os, shutil, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.getcwd())
import numpy as np
import bench, torch
from skder_amd import engine, synth
from skder_amd.skder import Database, lowMemGreedyDerep
N = int(os.environ.get("N", "20000"))
GZ = os.environ.get("GZ") == "1"
ctx = engine.Context(0)
recipe = synth.make_recipe(N, genome_len=2_800_000)
tmp = tempfile.mkdtemp(prefix="skder_amd_lowmem_files_")
paths, nbytes = [], 0
t0 = time.perf_counter()
try:
    for b0 in range(0, N, 1250):
        gs = range(b0, min(b0 + 1250, N))
        layout = engine.BatchLayout([recipe.rec_lens[g] for g in gs])
        d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
        ctx.synth_fill(d.data_ptr(), layout, recipe.lineage[gs.start:gs.stop], recipe.params[gs.start:gs.stop])
        sub, ps, nb = bench.write_sample_files([(layout, d)], len(gs))
        del d
        dst = os.path.join(tmp, "b%05d" % b0)
        os.rename(sub, dst)
        ps = [os.path.join(dst, os.path.basename(p)) for p in ps]
        if GZ:
            def z(p):
                with open(p, "rb") as f, gzip.open(p + ".gz", "wb", compresslevel=1) as g:
                    shutil.copyfileobj(f, g)
                os.remove(p)
                return p + ".gz"
            with ThreadPoolExecutor(max_workers=16) as ex:
                ps = list(ex.map(z, ps))
        paths += ps
        nbytes += nb
    t_write = time.perf_counter() - t0
    listing, n50_file = os.path.join(tmp, "listing.txt"), os.path.join(tmp, "Concatenated_N50.txt")
    open(listing, "w").write("".join(p + "\n" for p in paths))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    db = Database.from_listing(listing, n50_file=n50_file, device=0)
    t_ingest = time.perf_counter() - t0
    ws = os.path.join(tmp, "ws") + "/"
    os.makedirs(ws)
    res = os.path.join(ws, "skDER_Results.txt")
    t0 = time.perf_counter()
    lowMemGreedyDerep(listing, ws, n50_file, res, ws, 99.5, 50.0, None, database=db)
    t_sel = time.perf_counter() - t0
    reps = open(res).read().split()
    db.close()
    print(json.dumps({"genomes": N, "format": "fasta.gz (level 1)" if GZ else "plain FASTA", "fasta_text_bytes": nbytes,
                      "files_written_s": t_write, "ingest_s(listing -> resident database + N50 table)": t_ingest,
                      "ingest_text_GB_per_s": nbytes / t_ingest / 1e9, "low_mem_greedy_s(speculative search batches)": t_sel,
                      "files_to_listing_s": t_ingest + t_sel, "representatives": len(reps),
                      "reference_published": "README.md:27: > 20,000 Staphylococcus genomes, low_mem_greedy, 2.25 h on 20 threads (different data and hardware)"}))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
