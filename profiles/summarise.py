#!/usr/bin/env python3
"""profiles/summarise.py ROUND -- condense gpurun_out/prof_ROUND (written by profiles/collect.sh) into
profiles/ROUND_kernel_stats.csv, ROUND_domain_stats.csv, ROUND_pmc_calibration.json,
ROUND_pmc_traffic.json and ROUND_bench_line.json.

Counters: rocprofv3's FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch and summed over the
dispatches of one bench step (--steps 1 --warmup 0).  MI355X_MICROARCH.md (HBM section) says the
absolute values are uncalibrated on gfx950 except that a wide coalesced read reports one half of its
bytes, and prescribes a calibration on known byte counts in the kernel's own access pattern: that is
profiles/calib/fetch_calib.hip; the factor known_bytes / counter_bytes of the matching pattern is
applied to every kernel (`pattern` below says which one was used)."""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "round1"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + R)
DST = os.path.join(ROOT, "profiles")

# which calibration pattern describes the dominant stream of a kernel (reads, writes)
PATTERN = {
    "sketch_tiles_kernel": ("read16_coalesced", "write4_coalesced"),
    "join_probe_kernel": ("read4_coalesced", "write4_coalesced"),
    "chain_fast_kernel": ("read_line_per_lane", "write4_coalesced"),            # round 1's chaining kernel
    "run_extract_kernel": ("read16_coalesced", "write4_coalesced"),
    "chain_single_kernel": ("read_line_per_lane", "write4_coalesced"),
    "chain_runs_kernel": ("read_line_per_lane", "write4_coalesced"),
    # round 6: the index build's reads are the coalesced pass over the position-ordered seeds and, per seed, one 16-byte gather of its
    # (k-mer, position, record) copy through the bucket permutation -- the gather dominates the request count
    "index_genome_lds_kernel": ("gather16_random", "write4_coalesced"),
}
# kernels without a pattern of their own (copies, fills, scans, table builds: streaming accesses): the guide's rule for gfx950 -- a wide
# coalesced read reports half its bytes -- i.e. the read16_coalesced factor for FETCH_SIZE, WRITE_SIZE as counted
DEFAULT_PATTERN = ("read16_coalesced", "write4_coalesced")


def counters(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = re.sub(r"<.*>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").strip()     # (template kernels: "void name<args>(...)")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def find(d, suffix):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    raise FileNotFoundError(suffix + " under " + d)


def main():
    shutil.copy(find(os.path.join(SRC, "stats"), "kernel_stats.csv"), os.path.join(DST, R + "_kernel_stats.csv"))
    shutil.copy(find(os.path.join(SRC, "stats"), "domain_stats.csv"), os.path.join(DST, R + "_domain_stats.csv"))
    if os.path.isdir(os.path.join(SRC, "stats1q")):     # the same command with SKDER_AMD_QUEUES=1 (no overlap between chaining batches)
        shutil.copy(find(os.path.join(SRC, "stats1q"), "kernel_stats.csv"), os.path.join(DST, R + "_kernel_stats_one_queue.csv"))
    # SQ counters of the longest kernels (collect.sh runs profiles/tools/pmc.sh for each): kept per round next to the statistics
    for f in sorted(os.listdir(SRC)):
        if f.startswith("pmc_") and f.endswith(".txt") and os.path.getsize(os.path.join(SRC, f)) > 0:
            shutil.copy(os.path.join(SRC, f), os.path.join(DST, R + "_" + f))
    known = {}
    for line in open(os.path.join(SRC, "calib_fetch.log")):
        if line.startswith("known_bytes"):
            _, name, b = line.split()
            known[name] = int(b)
    cf = counters(find(os.path.join(SRC, "calib_fetch"), "counter_collection.csv"))
    cw = counters(find(os.path.join(SRC, "calib_write"), "counter_collection.csv"))
    calib = {}
    for name, b in known.items():
        src = cw if name.startswith("write") else cf
        kb = src[name][1]
        calib[name] = {"known_bytes": b, "counter_kb": kb, "counter_bytes": kb * 1024.0,
                       "factor": b / (kb * 1024.0) if kb else None}
    json.dump(calib, open(os.path.join(DST, R + "_pmc_calibration.json"), "w"), indent=1)
    f = counters(find(os.path.join(SRC, "fetch"), "counter_collection.csv"))
    w = counters(find(os.path.join(SRC, "write"), "counter_collection.csv"))
    traffic = {}
    for k in sorted(set(f) | set(w)):
        e = {}
        for cname, src, idx in (("FETCH_SIZE", f, 0), ("WRITE_SIZE", w, 1)):
            n, kb = src.get(k, [0, 0.0])
            e[cname] = {"launches": n, "sum_counter_kb": kb, "bytes_per_launch": kb * 1024.0 / n if n else 0.0}
            pat = PATTERN.get(k, (None, None))[idx]
            if pat and calib.get(pat, {}).get("factor"):
                e[cname]["pattern"] = pat
                e[cname]["factor"] = calib[pat]["factor"]
                e[cname]["corrected_bytes"] = kb * 1024.0 * calib[pat]["factor"]
        traffic[k] = e
    # the whole step: every kernel's counters after the calibration factors (its own pattern, else DEFAULT_PATTERN); the input generator
    # (synth_fill_kernel) is not part of a step
    step = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "kernels": 0}
    for k, e in traffic.items():
        if k.startswith("synth_fill"):
            continue
        step["kernels"] += 1
        for idx, cname in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
            fac = calib.get(DEFAULT_PATTERN[idx], {}).get("factor") or 1.0
            step[cname] += e[cname].get("corrected_bytes", e[cname]["sum_counter_kb"] * 1024.0 * fac)
    step["bytes"] = step["FETCH_SIZE"] + step["WRITE_SIZE"]
    step["what"] = ("sum over the kernels of one step (--steps 1 --warmup 0) of FETCH_SIZE and WRITE_SIZE after the calibration factors: "
                    "a kernel's own pattern where it has one, else the guide's rule for streaming reads (x %.3f) and writes as counted"
                    % (calib.get(DEFAULT_PATTERN[0], {}).get("factor") or 1.0))
    traffic["__step__"] = step
    json.dump(traffic, open(os.path.join(DST, R + "_pmc_traffic.json"), "w"), indent=1)
    line = json.loads([l for l in open(os.path.join(SRC, "bench_line.json")) if l.startswith("{")][-1])
    # the bench line was printed before this summary existed: put this collection's traffic into it
    dom = line["roofline"]["kernel"].split("+")[0]
    if dom in traffic:
        line["roofline"]["traffic"] = sum(traffic[dom][c].get("corrected_bytes", traffic[dom][c]["sum_counter_kb"] * 1024.0)
                                          for c in ("FETCH_SIZE", "WRITE_SIZE"))
    json.dump(line, open(os.path.join(DST, R + "_bench_line.json"), "w"), indent=1)
    for k in PATTERN:
        if k in traffic:
            t = traffic[k]
            print(k, "FETCH raw %.2f GB corrected %.2f GB; WRITE raw %.2f GB corrected %.2f GB" % (
                t["FETCH_SIZE"]["sum_counter_kb"] / 1048576, t["FETCH_SIZE"].get("corrected_bytes", 0) / 2**30,
                t["WRITE_SIZE"]["sum_counter_kb"] / 1048576, t["WRITE_SIZE"].get("corrected_bytes", 0) / 2**30))
    print(json.dumps(calib, indent=1))


if __name__ == "__main__":
    main()
