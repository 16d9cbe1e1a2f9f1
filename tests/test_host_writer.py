"""The edge-table side of the drop-in on the CPU (no GPU needed: pure host code of skder_amd/csrc/host_io.hip): row order
established in place, parallel sort of search tables, blocks of text formatted in parallel and written at their offsets --
each against the simple single-threaded statement, under AddressSanitizer + UBSan (tests/host_writer_harness.cpp)."""
import json
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "skder_amd", "csrc")


def _build(tmp_path, sanitize):
    """sanitize: False (optimised), True (AddressSanitizer + UBSan) or "thread" (ThreadSanitizer)"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)
    # every object of the library except host_io.o, which the harness compiles from source under the sanitizers (list: csrc/Makefile)
    mk = open(os.path.join(CSRC, "Makefile")).read()
    srcs = re.search(r"^SRC = (.*)$", mk, re.M).group(1).split()
    objs = [os.path.join(CSRC, f[:-4] + ".o") for f in srcs if f != "host_io.hip"] + [os.path.join(CSRC, o) for o in ("gunzip.o", "select.o")]
    exe = str(tmp_path / ("writer_harness" + ("_tsan" if sanitize == "thread" else "_san" if sanitize else "")))
    flags = (["-O1", "-g", "-Xarch_host", "-fsanitize=thread"] if sanitize == "thread" else
             ["-O1", "-g", "-Xarch_host", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if sanitize else ["-O3"])
    subprocess.check_call([hipcc, "--offload-arch=gfx950"] + flags + ["-std=c++17", "-I" + CSRC, "-x", "hip",
                           os.path.join(ROOT, "tests", "host_writer_harness.cpp"), os.path.join(CSRC, "host_io.hip"), "-x", "none"] + objs +
                          ["-o", exe, "-lz", "-lpthread"], stderr=subprocess.DEVNULL)
    return exe


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_row_order_and_table_text_under_sanitizers(tmp_path):
    exe = _build(tmp_path, True)
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", SKDER_AMD_IO_THREADS="6"))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-3000:]
    assert "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_two_million_rows_are_ordered_and_written(tmp_path):
    """a fifth of BASELINE config 5's table on the build container's 8 cores (profiles/ holds the 10^7-row timing)"""
    exe = _build(tmp_path, False)
    out = subprocess.run([exe, str(tmp_path), "2000000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(out.stdout.splitlines()[0])
    assert r["rows_kept"] > 1_000_000 and r["table_bytes"] > 150 * r["rows_kept"]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_row_order_and_writer_under_thread_sanitizer(tmp_path):
    """the same harness under ThreadSanitizer: the parallel row orders (histograms, index scatter, window jobs, gather, the spare list
    behind its mutex) and the block writer's hand-over of offsets run on several threads without a data race"""
    exe = _build(tmp_path, "thread")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=1500,
                         env=dict(os.environ, SKDER_AMD_IO_THREADS="6", TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0"))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-3000:]
    assert "ThreadSanitizer" not in out.stderr, out.stderr[-4000:]
