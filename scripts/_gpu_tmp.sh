timeout 300 python -m pytest tests -m gpu -q -x -k "stochastic or sharding_paths or rng" 2>&1 | tail -3
timeout 300 python bench.py --workload mc_evolved --steps 5 --warmup 2 --no-also --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], d['config']['price'])"
timeout 300 python bench.py --workload mc --steps 5 --warmup 2 --no-also --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], d['config']['price'])"
