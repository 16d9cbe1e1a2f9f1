"""round 4: low_mem_greedy on N genomes of ONE species FROM .fasta.gz FILES (the reference's inputs are .fasta.gz): the 34 real assemblies +
device-generated descendants written as gzip files (level 1), then listing -> ingest (read, inflate, PCIe, device parse, N50, sketch, index)
-> lowMemGreedyDerep -i 99.5 -f 50 with the database resident.  N=20000 by default; 5000 when the box has less than 40 GB of scratch."""
import gzip, json, os, shutil, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.getcwd())
import numpy as np
import bench, torch
from skder_amd import engine, synth
from skder_amd.skder import Database, lowMemGreedyDerep
N = int(os.environ.get("N", "20000"))
tmp = tempfile.mkdtemp(prefix="skder_amd_one_species_files_")
if shutil.disk_usage(tmp).free < 40e9 and N > 5000:
    N = 5000
ctx = engine.Context(0)
paths, nbytes, zbytes = [], 0, 0
pool = ThreadPoolExecutor(max_workers=16)


def z(p):
    with open(p, "rb") as f, gzip.open(p + ".gz", "wb", compresslevel=1) as g:
        shutil.copyfileobj(f, g)
    os.remove(p)
    return p + ".gz"


def keep(batch, d, layout):
    global nbytes, zbytes
    sub, ps, nb = bench.write_sample_files([(layout, d)], layout.n_genomes)
    dst = os.path.join(tmp, "b%05d" % batch)
    os.rename(sub, dst)
    ps = list(pool.map(z, [os.path.join(dst, os.path.basename(p)) for p in ps]))
    paths.extend(ps)
    nbytes += nb
    zbytes += sum(os.path.getsize(p) for p in ps)


try:
    t0 = time.perf_counter()
    sk, names, n50, _ = bench.one_species_database(engine, ctx, torch, synth, N, 0, keep_bases=keep)
    sk.close()
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
    st = dict(getattr(lowMemGreedyDerep, "last_stats", {}) or {})
    reps = open(res).read().split()
    db.close()
    print(json.dumps({"genomes": len(paths), "format": "fasta.gz (level 1)", "fasta_text_bytes": nbytes, "gz_bytes": zbytes,
                      "files_written_s": t_write, "ingest_s(listing -> resident database + N50 table)": t_ingest,
                      "ingest_text_GB_per_s": nbytes / t_ingest / 1e9, "low_mem_greedy_s(speculative search batches)": t_sel,
                      "files_to_listing_s": t_ingest + t_sel, "representatives": len(reps), "search_stats": st,
                      "workload": "ONE species: the reference's 34 real C. granulosum assemblies + descendants generated on the device (descend.hip), "
                                  "every genome within ANI range of every other",
                      "reference_published": "README.md:27: > 20,000 Staphylococcus genomes, low_mem_greedy, 2.25 h on 20 threads (different data and hardware)"}))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
