export RMHIP_BENCH_BACKEND=gloo
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo rc=$?; tail -5 gpurun_out/bench_2rank.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_2rank.json').read().strip().splitlines()[-1])
print(d['n_gpus'], d['metric'], d['value'], d['ms_per_step'], d['scaling'])
for a in d.get('also', []): print(a['metric'][:60], a['value'], a['ms_per_step'], a['config'].get('parallelism'), a['config'].get('price'))
PY
