export RMHIP_BENCH_BACKEND=gloo
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --workload mldivide --steps 1 --warmup 1 --no-also --no-cpu-baseline > gpurun_out/bench_2rank_lu.json 2> gpurun_out/bench_2rank_lu.err
echo rc=$?; grep -v "socket.cpp\|amdgpu.ids" gpurun_out/bench_2rank_lu.err | tail -8 | cut -c1-300
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_2rank_lu.json').read().strip().splitlines()[-1])
print(d['n_gpus'], d['metric'], d['value'], d['ms_per_step'], d['config'])
PY
