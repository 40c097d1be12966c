mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.err
RMHIP_BENCH_FORCE_CYCLIC=1 timeout 300 python bench.py --workload mldivide --steps 3 --warmup 1 --no-also --no-cpu-baseline > gpurun_out/bench_cyclic.json 2>&1
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'])
for a in d.get('also', []): print(a['metric'], a['value'], a['ms_per_step'], a.get('roofline', {}).get('frac'))
c = json.loads(open('gpurun_out/bench_cyclic.json').read().strip().splitlines()[-1])
print('cyclic:', c['value'], c['ms_per_step'], c['config'])
PY
