#!/bin/bash
# development probe: P8 GEMM variants (ECAMP_P8_DBG bits: 4 = no epilogue, 8 = no pre-epilogue DMA wait) vs the 128^2 kernel
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for dbg in ${DBGS:-0}; do
  echo "== P8 forced, ECAMP_P8_DBG=$dbg"
  ECAMP_GEMM_P8=2 ECAMP_P8_DBG=$dbg timeout 200 python tools/gemm_bench.py $SHAPES 2>&1 | grep -v amdgpu.ids | tail -15
done
echo "== 128^2 kernel only"
ECAMP_GEMM_P8=0 timeout 200 python tools/gemm_bench.py $SHAPES 2>&1 | grep -v amdgpu.ids | tail -15
