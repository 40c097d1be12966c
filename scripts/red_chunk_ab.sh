#!/bin/bash
# Developer tool (GPU box): sum(x,2) against the number of columns per chunk of kernel B (RMHIP_RED_B_CHUNK).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for ch in 0 96 112 120 125 127 128 129 136 144 160 192 255 256; do
  echo -n "chunk=$ch: "; RMHIP_RED_B_CHUNK=$ch timeout 100 python scripts/red_shapes.py 2>&1 | grep "(8192, 8192)\|(16384, 4096)" | sed 's/| sum(x,1).*//' | tr '\n' ' '; echo
done
