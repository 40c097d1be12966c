#!/bin/bash
# Developer tool (GPU box): SQ / TCC counters of the rank-k update kernel at the LU's shapes against a deep product (scripts/lu_update_case.py),
# one block per CU (beta = 1: the C-preloading variant, 138 VGPRs) and two (beta = 0).  Separate --pmc passes, kernel trace only.
# Usage: scripts/lu_update_pmc.sh   -> gpurun_out/lu_update_pmc.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lu_update_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for CASE in "rank512 1" "rank512 0" "rank128 1" "deep 1" "deep 0"; do
  T=$(echo $CASE | tr ' ' '_')
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/${T}_p$i" -o p -- python $ROOT/scripts/lu_update_case.py $CASE > "$OUT/${T}_p$i.log" 2>&1
  done
done
python - "$OUT" <<'PY' > $ROOT/gpurun_out/lu_update_pmc.txt
import collections, csv, glob, sys, os
out = sys.argv[1]
for case in ("rank512_1", "rank512_0", "rank128_1", "deep_1", "deep_0"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{case}_p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_dgemm" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    logs = [open(f).read().strip().splitlines()[-1] for f in sorted(glob.glob(f"{out}/{case}_p1.log"))]
    print("==", case, logs)
    for k in sorted(agg):
        v = agg[k]
        print(f"   {k:34s} n={len(v):2d} mean={sum(v)/len(v):16.1f}")
PY
cat $ROOT/gpurun_out/lu_update_pmc.txt
