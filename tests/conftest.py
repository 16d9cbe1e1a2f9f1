import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure); builds it on first use"""
    import subprocess
    if not os.path.isfile(os.path.join(ROOT, "oracle", "libani_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    import oracle_py
    return oracle_py


@pytest.fixture(scope="session")
def ref_bins():
    """the reference's own skDERsum / skDERcore (oracle/_ref), compiled from the sources under /root/reference by oracle/build_ref.sh
    when that tree exists; None where it does not (the GPU box) -- tests then use committed outputs / digests of these binaries"""
    import subprocess
    d = os.path.join(ROOT, "oracle", "_ref")
    have = lambda: all(os.path.isfile(os.path.join(d, b)) for b in ("skDERsum", "skDERcore"))
    if not have() and os.path.isdir("/root/reference"):
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")], stdout=subprocess.DEVNULL)
    return {b: os.path.join(d, b) for b in ("skDERsum", "skDERcore")} if have() else None


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_table(path):
    rows = []
    with open(path) as f:
        header = next(f)
        for line in f:
            rows.append(line.rstrip("\n").split("\t"))
    return header, rows
