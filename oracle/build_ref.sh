#!/bin/bash
# Compiles the reference's two native helpers (the DOWN-STREAM consumers of the skani edge table) from
# their sources where they lie under /root/reference, outputs only into oracle/_ref/ (git-ignored).
# Used by tests/test_selection.py to validate skder_amd/selection.py against the real binaries.
# The hot path itself (skani) is a third-party Rust program that is NOT in the reference tree and
# cannot be built here (no Rust toolchain): unbuildable, see DESIGN.md section 2.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference/src/skDER
[ -d "$REF" ] || { echo "reference not present; skipping"; exit 0; }
mkdir -p "$HERE/_ref"
g++ -O2 -std=c++11 -o "$HERE/_ref/skDERsum" "$REF/skDERsum.cpp"
g++ -O2 -std=c++11 -o "$HERE/_ref/skDERcore" "$REF/skDERcore.cpp"
echo "built $HERE/_ref/skDERsum $HERE/_ref/skDERcore"
