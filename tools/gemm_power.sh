#!/bin/bash
# Development probe: socket power and shader clock while ONE GEMM shape runs back to back, random against all-zero operands
# (is the K loop's zero-vs-random gap the power limit?).   usage (GPU box): bash tools/gemm_power.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/tools/gemm_lab
for mode in "" "--zero"; do
  ./lab $mode --n=120000 --sch=1 --quick --forms=1 sq4k > /tmp/gp_$$.log 2>&1 &
  LP=$!
  sleep 4
  for i in 1 2 3 4; do
    echo -n "operands ${mode:-random}: "; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket|sclk" | sed -E 's/GPU\[0\]\s*: //' | tr '\n' ';' ; echo
    sleep 1
  done
  wait $LP
  grep "plain" /tmp/gp_$$.log | cut -c1-110
done
