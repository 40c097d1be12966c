#!/bin/bash
# Developer tool (GPU box): the round-5 evidence run - the un-profiled default bench line, then scripts/profile_r05.sh for the solve
# (WORKLOADS=mldivide); copy into profiles/ with scripts/collect_profiles.py r05.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r05
python bench.py > gpurun_out/prof_r05/bench_default.json 2> gpurun_out/prof_r05/bench_default.err
tail -c 400 gpurun_out/prof_r05/bench_default.json
WORKLOADS=mldivide bash scripts/profile_r05.sh > gpurun_out/prof_r05/profile.log 2>&1
tail -5 gpurun_out/prof_r05/profile.log
head -30 gpurun_out/prof_r05/lu_attribution.txt
