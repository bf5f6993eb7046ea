#!/bin/bash
# L2 (TCC) hit / miss / fabric-read counters of the GEMM kernels on three forward shapes (one --pmc group per pass).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/gemm_l2/$tag -- python3 $R/tools/gemm_bench.py "enc fc1" "bert inter" "dec fc2" > $R/gpurun_out/gemm_l2.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/gemm_l2 | grep "gemm_bf16\|^kernel"
