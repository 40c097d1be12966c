#!/bin/bash
# interleaved A/B of k_rng_normal / the Monte-Carlo step over library builds: scripts/rng_ab.sh <lib>...   ("cur" = the in-tree build)
for round in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = cur ]; then unset RMHIP_LIBRARY; else export RMHIP_LIBRARY=$PWD/ab_old/$lib/librmhip.so; fi
    echo "== $lib (round $round)"; python scripts/rng_time.py 40 2>&1 | grep -v amdgpu.ids
  done
done
