#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab.py --rounds 4 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/c3par.so variants/c3np.so dsp_amd/libdsp_amd.so > gpurun_out/r3/ab_c3_np.txt 2>&1
tail -3 gpurun_out/r3/ab_c3_np.txt
for lib in variants/c3np.so dsp_amd/libdsp_amd.so; do
  tag=$(basename $lib .so)
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH"; do
    DSP_AMD_LIB=$GRAFT_REPO_ROOT/$lib bash tools/pmc_any.sh c3_${tag}_$(echo $grp | cut -d' ' -f1) mfcc1024 "$grp" -- --workload config3 --no-config4 --settle 0 2>&1 | tail -1
  done
done > gpurun_out/r3/pmc_c3_compare.txt 2>&1
cat gpurun_out/r3/pmc_c3_compare.txt
