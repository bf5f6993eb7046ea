#!/bin/bash
# the step beside a co-tenant that holds CUs during ~10 ms of backward (a stand-in for RCCL's all-reduce workgroups), by launch policy:
# weight-gradient reserve (P8_RESERVE) x data-gradient grid (Q8_BWD_GRID: 0 persistent, large = one tile per workgroup)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "0 0" "32 0" "32 1048576"; do set -- $cfg; HOG_THREADS=${HOG_THREADS:-1024} P8_RESERVE=$1 Q8_BWD_GRID=$2 timeout 300 python tools/hog_probe.py bwd 2>&1 | grep -v amdgpu.ids; done
