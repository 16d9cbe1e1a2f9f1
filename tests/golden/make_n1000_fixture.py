"""Digests of the REFERENCE's own skDERsum / skDERcore on the 1,000-genome table of tests/test_gpu_parity.py::
test_driver_greedy_and_dynamic_on_1000_genomes, for boxes without /root/reference (the GPU box).

    1. on the GPU box:  SKDER_AMD_DUMP_N1000=gpurun_out/n1000 python -m pytest tests/test_gpu_parity.py -m gpu -k 1000_genomes
       (writes the table and the N50 file with the scratch directory replaced by /G/)
    2. in the build container (oracle/_ref built from /root/reference by oracle/build_ref.sh):
       python tests/golden/make_n1000_fixture.py gpurun_out/n1000

Only digests are committed: the member lists of 1,000 genomes are 6 MB of text."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = sys.argv[1]
table, n50 = os.path.join(d, "table.tsv"), os.path.join(d, "n50.txt")
ref = os.path.join(ROOT, "oracle", "_ref")
run = lambda *a: subprocess.run(list(a), capture_output=True, text=True, check=True).stdout
sha = lambda t: hashlib.sha256(t.encode()).hexdigest()
out = {"table_sha256": sha(open(table).read()),
       "skDERsum_98.5_50_sha256": sha(run(os.path.join(ref, "skDERsum"), table, n50, "98.5", "50.0")),
       "skDERcore_98.5_50_10_sha256": sha(run(os.path.join(ref, "skDERcore"), table, n50, "98.5", "50.0", "10.0")),
       "rows": sum(1 for _ in open(table)) - 1,
       "made_by": "tests/golden/make_n1000_fixture.py; reference binaries compiled from /root/reference/src/skDER/skDERsum.cpp, skDERcore.cpp"}
with open(os.path.join(ROOT, "tests", "golden", "downstream", "n1000_reference_digests.json"), "w") as f:
    json.dump(out, f, indent=1)
print(out)
