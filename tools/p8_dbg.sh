#!/bin/bash
# development probe: where does the P8 GEMM spend its time? (dbg: 1 no in-loop DMA, 2 no reads/MFMA, 4 no epilogue, 8/16 DMA placement)
cd /root/repo
for dbg in ${DBGS:-0 4}; do
  echo "== ECAMP_P8_DBG=$dbg"
  ECAMP_GEMM_P8=${P8MODE:-2} ECAMP_P8_DBG=$dbg timeout 200 python tools/gemm_bench.py $SHAPES 2>&1 | grep -v amdgpu.ids | tail -15
done
echo "== baseline (128^2 kernel)"
timeout 200 python tools/gemm_bench.py $SHAPES 2>&1 | grep -v amdgpu.ids | tail -15
