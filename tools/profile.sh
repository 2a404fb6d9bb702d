#!/bin/bash
# tools/profile.sh TAG -- rocprofv3 passes for bench.py on the GPU box.
# Kernel trace + stats in one pass; each PMC group in its own pass (never combined
# with sys/hip/hsa traces).  Outputs under gpurun_out/prof_TAG/.
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 200 --warmup 20 > "$OUT/trace.json" 2> "$OUT/trace.err"
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- $BENCH --steps 4 --warmup 2 --settle 0 > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pmc pass $name failed" >> "$OUT/errors.txt"
done
find "$OUT" -name "*.csv" | head -50
