"""CPU tests of oracle/skani_ref.py, the probe for the real `skani` binary (BASELINE.md plan item 1).

skani is absent from the build container and from the GPU boxes used so far, so the probe's two states are covered here:
absent (the callers must say so and fall back to the repo's own restatement, labelled "port") and present -- with a
TEST DOUBLE on PATH: a shell script that answers `-V` and writes the reference's own golden table for `triangle`.  The
double exercises the command line, the table parser and the cell-by-cell comparison; it is not a reference build and
nothing outside this test ever runs it."""
import os
import stat
import sys

import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_absent_skani_is_reported_not_replaced(monkeypatch, tmp_path):
    import skani_ref
    monkeypatch.setenv("PATH", str(tmp_path))            # nothing there
    assert skani_ref.find() is None and skani_ref.version() is None
    assert skani_ref.triangle(str(tmp_path / "l.txt"), str(tmp_path / "o.tsv"), 50.0, 80.0, 4) is None
    assert not (tmp_path / "o.tsv").exists()


def _install_double(tmp_path, monkeypatch, golden_table, fail=False):
    exe = tmp_path / "bin" / "skani"
    exe.parent.mkdir()
    log = tmp_path / "argv.log"
    body = "#!/bin/sh\nif [ \"$1\" = \"-V\" ]; then echo 'skani 0.0.0-test-double'; exit 0; fi\necho \"$@\" > %s\n" % log
    if not fail:
        body += "out=''\nwhile [ $# -gt 0 ]; do if [ \"$1\" = \"-o\" ]; then out=\"$2\"; fi; shift; done\ncp %s \"$out\"\n" % golden_table
    exe.write_text(body)
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", str(exe.parent) + os.pathsep + "/usr/bin:/bin")
    return log


def test_present_skani_is_run_with_the_reference_command_line_and_compared(monkeypatch, tmp_path):
    import skani_ref
    g5 = os.path.join(GOLDEN, "G5_triangle_minaf10_s89.5.tsv")
    log = _install_double(tmp_path, monkeypatch, g5)
    assert skani_ref.version() == "skani 0.0.0-test-double"
    out = tmp_path / "skani.tsv"
    run = skani_ref.triangle("LISTING", str(out), 10.0, 89.5, 4)
    # skder.py:16-18: skani triangle -l L --min-af A -E <params> -t T -o OUT
    assert log.read_text().split() == ["triangle", "-l", "LISTING", "--min-af", "10.0", "-E", "-s", "89.5", "-t", "4", "-o", str(out)]
    assert run["rows"] == 561 and run["version"].startswith("skani") and run["seconds"] >= 0
    same = skani_ref.compare_tables(str(out), g5)
    assert same["pairs"] == 561 and same["max_abs_dANI"] == 0 and same["max_abs_dAF"] == 0
    assert same["identical_ANI_cells"] == 561 and same["identical_AF_cells"] == 1122 and same["only_mine"] == same["only_theirs"] == 0
    # a table with the roles of one pair swapped and one value moved: AF columns follow the roles, the difference is found
    rows = open(g5).read().splitlines()
    c = rows[1].split("\t")
    c[0], c[1], c[3], c[4], c[5], c[6] = c[1], c[0], c[4], c[3], c[6], c[5]
    c[2] = "%.2f" % (float(c[2]) - 0.25)
    other = tmp_path / "other.tsv"
    other.write_text("\n".join([rows[0], "\t".join(c)] + rows[2:-1]) + "\n")
    d = skani_ref.compare_tables(str(other), g5)
    assert d["pairs"] == 560 and d["only_theirs"] == 1 and abs(d["max_abs_dANI"] - 0.25) < 1e-9 and d["max_abs_dAF"] == 0


def test_skani_that_writes_no_table_is_a_failure(monkeypatch, tmp_path):
    """util.runCmd's rule (util.py:636-652): success == the output file exists"""
    import skani_ref
    _install_double(tmp_path, monkeypatch, "", fail=True)
    with pytest.raises(RuntimeError, match="Had an issue running"):
        skani_ref.triangle("LISTING", str(tmp_path / "none.tsv"), 50.0, 80.0, 2)


def test_probe_looks_beyond_path_in_conda_environments(monkeypatch, tmp_path):
    """skDER is installed from bioconda (skDER_env.yml:12): a box may hold skani in an environment that is not activated"""
    import skani_ref
    g5 = os.path.join(GOLDEN, "G5_triangle_minaf10_s89.5.tsv")
    _install_double(tmp_path, monkeypatch, g5)
    env_bin = tmp_path / "conda" / "bin"
    env_bin.parent.mkdir()
    os.rename(tmp_path / "bin", env_bin)
    monkeypatch.setenv("PATH", "/usr/bin:/bin")
    monkeypatch.setenv("HOME", str(tmp_path / "nobody"))
    monkeypatch.delenv("CONDA_PREFIX", raising=False)
    monkeypatch.delenv("MAMBA_ROOT_PREFIX", raising=False)
    if skani_ref.find() is None:                 # (a machine that really has skani somewhere is not this test's business)
        monkeypatch.setenv("CONDA_PREFIX", str(tmp_path / "conda"))
        assert skani_ref.find() == str(env_bin / "skani") and skani_ref.version() == "skani 0.0.0-test-double"
        monkeypatch.setenv("SKANI_REF_NO_SEARCH", "1")
        assert skani_ref.find() is None
