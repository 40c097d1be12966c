#!/bin/bash
# Developer tool (GPU box): rank-k update rates with one vs two dgemm blocks per CU.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SHAPES="16128:15872:256 12288:12288:256 8192:8192:256 4096:4096:256 15872:15360:512 8192:8192:512 8192:8192:128 16128:256:256 8192:256:256 4096:256:256 2048:256:256 16256:128:128 8192:128:128 16320:64:64 8192:64:64 4096:64:64"
for pad in 0 10752; do
  echo "== RMHIP_GEMM_LDS_PAD=$pad"
  RMHIP_GEMM_LDS_PAD=$pad python $ROOT/scripts/gemm_shapes.py $SHAPES
done
