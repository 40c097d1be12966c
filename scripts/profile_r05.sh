#!/bin/bash
# Developer tool (GPU box): round-5 evidence.  (1) rocprofv3 kernel-trace stats + counter passes of BASELINE's workloads (profile_r04.sh's
# passes under the tag r05); (2) the solve's kernel trace -> per-stream timeline, per-launch update rates, main-stream statistics, schedule
# skeleton; (3) the attribution A/B of the round's switches (wall clock, interleaved); (4) the main stream's timed-event timeline.
# Usage: scripts/profile_r05.sh            results under gpurun_out/prof_r05/ (copy into profiles/ with scripts/collect_profiles.py r05)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r05
mkdir -p "$OUT"
export RMHIP_BENCH_BUSY_S=0
bash $ROOT/scripts/profile_r04.sh r05 ${WORKLOADS:-fused dgemm mc chain mldivide}
bash $ROOT/scripts/lu_super_trace.sh r05trace 16384
T=$ROOT/gpurun_out/lu_r05trace
{ echo "== x = A\\b, n = 16384, two-level driver: rocprofv3 --kernel-trace of scripts/lu_trace.py (2 solves), last solve"; awk '/== last solve/,0' $T/timeline.txt | cut -c1-170;
  echo; echo "== per-launch update rates by stream (shapes from RMHIP_LU_GEMM_LOG matched against the trace)"; cat $T/gemm_rates.txt;
  echo; echo "== main stream, per 10 ms window: launches x mean duration"; python $ROOT/scripts/lu_main_stats.py $T;
  echo; echo "== schedule skeleton: kernels >= 400 us and main-stream idle gaps >= 200 us"; python $ROOT/scripts/lu_skeleton.py $T 400 200; } > "$OUT/mldivide_timeline.txt" 2>&1
{ echo "== wall clock of x = A\\b at n = 16384 (scripts/lu_trace.py, 4 solves each: first is cold), interleaved on one box";
  bash $ROOT/scripts/lu_super_ab.sh - RMHIP_LU_SUPER=0 RMHIP_LU_YIELD=0 RMHIP_LU_GEMM_PRIO=0 RMHIP_LU_RB_MFMA=0 RMHIP_LU_TRSM_MFMA=0 RMHIP_LU_RB_MFMA=0,RMHIP_LU_TRSM_MFMA=0,RMHIP_LU_YIELD=0,RMHIP_LU_GEMM_PRIO=0 RMHIP_LU_SUPER=0,RMHIP_LU_RB_MFMA=0,RMHIP_LU_TRSM_MFMA=0 - RMHIP_LU_SKIP=2 RMHIP_LU_SKIP=13 RMHIP_LU_SUPER_SEQ=512:512/1024:512/2048:512,RMHIP_LU_SUPER_ROWS=6144 RMHIP_LU_SUPER_SEQ=512:512 RMHIP_LU_SUPER_LATE=512:128 RMHIP_LU_FAR_PAD=0 RMHIP_LU_MID_PAD=0 -;
  for n in 12288 8192 4096; do echo "== n = $n"; N=$n bash $ROOT/scripts/lu_super_ab.sh - RMHIP_LU_SUPER=0,RMHIP_LU_RB_MFMA=0,RMHIP_LU_TRSM_MFMA=0; done; } > "$OUT/lu_attribution.txt" 2>&1
cd /tmp; RMHIP_LU_TIMELINE=1 python $ROOT/scripts/lu_trace.py 16384 2 2>&1 | grep -E "timeline|rep=" | awk '/rep=0/{f=1} f' | grep -v "rep=0" > "$OUT/mldivide_main_events.txt"
