cd $GRAFT_REPO_ROOT
run() { echo "== $1 N=$2"; env $1 timeout 200 python scripts/lu_stress.py $2 $3 | tail -4; }
run A=0 12288 40
run A=0 16384 15
run A=0 8192 30
run A=0 6144 30
run A=0 11264 20
run A=0 13312 20
