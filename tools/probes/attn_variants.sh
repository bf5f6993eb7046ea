#!/bin/bash
# builds tools/probes/attn_phase_probe.hip with each "-D..." variant given and runs case $CASE (default 1 = report side, hd 128)
R=${GRAFT_REPO_ROOT:-/root/repo}
CASE=${CASE:-1}
for v in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -DATTN_TS $v -I $R/ecamp_amd/csrc -o /tmp/attn_probe_v $R/tools/probes/attn_phase_probe.hip $R/ecamp_amd/csrc/core.hip > /tmp/attn_build.log 2>&1 || { echo "build failed: $v"; grep -E " error" -A3 /tmp/attn_build.log | head; continue; }
  echo "### $v"
  /tmp/attn_probe_v $CASE
done
