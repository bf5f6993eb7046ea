#!/bin/bash
# kernel table of one of the other configs: tools/config_kernels.sh <substring of the config name> <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_$2
ECAMP_OVERLAP_WGRAD=0 ECAMP_OVERLAP_BRANCHES=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$2 -o $2 -- python3 $R/tools/config_runs.py --steps 4 --only "$1" > $R/gpurun_out/prof_$2.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_$2/*.db | head -1) --skip-first-frac 0.65 > $R/gpurun_out/kernel_stats_$2.txt
rm -rf $R/gpurun_out/prof_$2
