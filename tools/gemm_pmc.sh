#!/bin/bash
# SQ counters of the GEMM kernels on two forward shapes (separate --pmc passes, nothing else traced):
#   gpurun -- 'bash tools/gemm_pmc.sh'   ->  gpurun_out/gemm_pmc/**/counter_collection.csv, summary on stdout
# WAVE_CYCLES / WAIT_* / ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES are cycles summed over SIMDs
# (= 16 per v_mfma_f32_16x16x32_bf16).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/gemm_pmc/$tag -- python3 $R/tools/gemm_bench.py "enc fc1" "bert inter" > $R/gpurun_out/gemm_pmc.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/gemm_pmc | grep "gemm_bf16\|^kernel"
