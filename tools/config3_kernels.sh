#!/bin/bash
# kernel table of configs[3] (ViT-L/16 at 448^2, B=64): tools/config3_kernels.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_c3
ECAMP_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c3 -o c3 -- python3 $R/tools/config_runs.py --steps 4 --only "configs[3]" > $R/gpurun_out/prof_c3.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_c3/*.db | head -1) --skip-first-frac 0.65 > $R/gpurun_out/kernel_stats_c3.txt
rm -rf $R/gpurun_out/prof_c3
