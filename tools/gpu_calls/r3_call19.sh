#!/bin/bash
mkdir -p gpurun_out/r3
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY"; do
  bash tools/pmc_any.sh c3lds_$(echo $grp | cut -d' ' -f1) mfcc1024 "$grp" -- --workload config3 --no-config4 --settle 0 2>&1 | tail -1
done > gpurun_out/r3/pmc_c3_lds.txt 2>&1
cat gpurun_out/r3/pmc_c3_lds.txt
