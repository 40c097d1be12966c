#!/bin/bash
# Developer tool (GPU box): interleaved A/B of the generated streaming kernel's tuning knobs on the headline workload (kernel ms by HIP events).
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "-" ] && envs="A=0"
  r=$(env $envs RMHIP_CACHE_DIR=/tmp/rmhip_cache_$$ python bench.py --workload ${W:-fused} --steps 100 --warmup 10 --no-also --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'] if d['roofline'].get('kernel_ms') else d['ms_per_step'], d['roofline']['frac'])")
  echo "$spec : $r"
done; done
