#!/bin/bash
# The round's committed evidence in one GPU call: serialized + production kernel tables, PMC traffic passes, default bench line.
#   gpurun -- 'bash tools/round_profiles.sh <tag>'   ->  gpurun_out/{kernel_stats_<tag>_serialized.txt, kernel_stats_<tag>_production.txt, <round>_pmc_*, bench_<tag>.json}
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
bash $R/tools/step_kernels.sh ${TAG}_ser 16
mv $R/gpurun_out/kernel_stats_${TAG}_ser.txt $R/gpurun_out/kernel_stats_${TAG}_serialized.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}_prod
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_prod -o prod -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-prof --only-value > $R/gpurun_out/prof_${TAG}_prod.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_${TAG}_prod/*.db | head -1) --skip-first-frac 0.4 > $R/gpurun_out/kernel_stats_${TAG}_production.txt
rm -rf $R/gpurun_out/prof_${TAG}_prod
cd $R
bash tools/pmc_traffic.sh > /dev/null 2>&1
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/bench_${TAG}.json
head -c 600 gpurun_out/bench_${TAG}.json; echo
head -4 gpurun_out/kernel_stats_${TAG}_production.txt
