cd /tmp
for n in 16384 12288 8192 6144 4096 2048 1024; do python $GRAFT_REPO_ROOT/scripts/lu_trace.py $n 4 2>&1 | tail -2 | tr '\n' ' '; echo; done
cd $GRAFT_REPO_ROOT && timeout 900 python -m pytest tests/test_gpu_lookahead.py tests/test_gpu_parity.py -x -q -m gpu -k "lu or mldivide or linsolve or mrdivide or solve or lookahead or substitution" 2>&1 | tail -5
