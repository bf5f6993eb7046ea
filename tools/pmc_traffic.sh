#!/bin/bash
# two separate --pmc passes of the serialized bench step (nothing else traced), then the per-launch GEMM traffic summary
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RN=${ROUND_TAG:-r06}   # file-name prefix of the round
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
ECAMP_OVERLAP_WGRAD=0 ECAMP_OVERLAP_BRANCHES=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/pmc_fetch.log 2>&1
ECAMP_OVERLAP_WGRAD=0 ECAMP_OVERLAP_BRANCHES=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write > $R/gpurun_out/${RN}_pmc_traffic.json
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_fetch | head -40 > $R/gpurun_out/${RN}_pmc_fetch_size.txt
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_write | head -40 > $R/gpurun_out/${RN}_pmc_write_size.txt
# keep the merge-back small: the raw csv files are large
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
head -c 1500 $R/gpurun_out/${RN}_pmc_traffic.json
