#!/bin/bash
# Developer tool: build librmhip.so as of a git revision into ab_old/<name>/librmhip.so (git-ignored; travels to the GPU box) for
# interleaved A/B runs: RMHIP_LIBRARY=ab_old/<name>/librmhip.so.   Usage: scripts/build_rev.sh <name> <git-rev>
set -e
NAME=$1; REV=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/ab_old/$NAME
rm -rf "$OUT" && mkdir -p "$OUT/tree"
git -C "$ROOT" archive "$REV" runmat_amd/csrc include | tar -x -C "$OUT/tree"
make -C "$OUT/tree/runmat_amd/csrc" -j8 >/dev/null
cp "$OUT/tree/runmat_amd/csrc/librmhip.so" "$OUT/librmhip.so"
rm -rf "$OUT/tree"
echo "built $OUT/librmhip.so from $REV"
