#!/bin/bash
# Serialized kernel trace of the bench step -> per-kernel table: [BENCH_ARGS="--dtype fp16"] tools/step_kernels.sh <tag> [steps]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; STEPS=${2:-16}
rm -rf $R/gpurun_out/prof_$TAG
ECAMP_OVERLAP_WGRAD=0 ECAMP_OVERLAP_BRANCHES=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-prof --only-value $BENCH_ARGS > $R/gpurun_out/prof_$TAG.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_$TAG/*.db | head -1) --skip-first-frac 0.3 > $R/gpurun_out/kernel_stats_$TAG.txt
rm -rf $R/gpurun_out/prof_$TAG
