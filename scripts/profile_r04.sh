#!/bin/bash
# Developer tool (GPU box): the round's rocprofv3 evidence - kernel-trace stats of every bench workload and SEPARATE counter passes
# (FETCH_SIZE, WRITE_SIZE; for the GEMMs also the matrix-pipe counters; for every workload the VALU counters behind the "valu" rooflines:
# SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE) - summarised into gpurun_out/prof_<tag>/.
# Usage: scripts/profile_r04.sh <tag> [workloads...]     default workloads: all
# Counter passes never carry another trace domain besides --kernel-trace, every rocprofv3 run sits under `timeout`.
set -u
TAG=${1:-r04}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
WORKLOADS=${*:-fused dgemm mc mc_evolved image chain bcast fft fused_f32 sgemm mldivide reductions}
for W in $WORKLOADS; do
  case $W in
    reductions) CMD="python $ROOT/scripts/red_driver.py 6" ;;
    mldivide)   CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --workload mldivide" ;;
    *)          CMD="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --workload $W" ;;
  esac
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$W" -o trace -- $CMD > "$OUT/trace_${W}_bench.json" 2> "$OUT/trace_$W.err"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_$W" -o fetch -- $CMD > "$OUT/fetch_${W}_bench.json" 2> "$OUT/fetch_$W.err"
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_$W" -o write -- $CMD > "$OUT/write_${W}_bench.json" 2> "$OUT/write_$W.err"
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_valu_$W" -o valu -- $CMD > "$OUT/valu_${W}_bench.json" 2> "$OUT/valu_$W.err"
  if [ "$W" = dgemm ] || [ "$W" = sgemm ]; then
    C="SQ_INSTS_VALU_MFMA_MOPS_F64"; [ "$W" = sgemm ] && C="SQ_INSTS_VALU_MFMA_MOPS_F32"
    timeout 400 rocprofv3 --pmc $C SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma_$W" -o mfma -- $CMD > "$OUT/mfma_${W}_bench.json" 2> "$OUT/mfma_$W.err"
  fi
  find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
done
python $ROOT/scripts/pmc_summary.py "$OUT" $WORKLOADS
