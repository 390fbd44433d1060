#!/bin/bash
# build the library of a git revision (default HEAD) into ab/libfigh_prev.so for same-box A/B runs (tools/ab_step.sh)
set -e
REF=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/figh_prev.XXXXXX)
git -C "$ROOT" archive "$REF" figaroh_plus_amd/csrc include | tar -x -C "$TMP"
make -s -C "$TMP/figaroh_plus_amd/csrc" -j6 OUT="$TMP/libfigh_prev.so" 2>&1 | grep -v "loop not unrolled\|pass-failed" || true
mkdir -p "$ROOT/ab"
cp "$TMP/libfigh_prev.so" "$ROOT/ab/libfigh_prev.so"
rm -rf "$TMP"
echo "ab/libfigh_prev.so <- $REF"
