#!/bin/bash
# SQ / LDS counters of the Q8 lab kernels (separate --pmc passes, nothing else traced):
#   gpurun -- 'bash tools/gemm_lab/pmc.sh enc_fc1 --forms=1 --dbg=0,4,6'   ->  gpurun_out/lab_pmc/**; summary on stdout
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/lab_pmc
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/lab_pmc/$tag -- $R/tools/gemm_lab/lab --quick "$@" > $R/gpurun_out/lab_pmc.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/lab_pmc | grep "q8\|^kernel"
