#!/bin/bash
# Development probe: socket power and shader clock while the training step runs (is the step power-limited?).
# usage (GPU box): bash tools/power_probe.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/bench.py --steps 1500 --warmup 10 --no-cpu-baseline --no-prof --only-value > /tmp/pp_bench.log 2>&1 &
BP=$!
sleep 20
for i in $(seq 1 16); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket|sclk" | sed -E 's/GPU\[0\]\s*: //' | tr '\n' ';' ; echo
  sleep 3
done
wait $BP
tail -1 /tmp/pp_bench.log | cut -c1-200
echo "idle:"; sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ';'; echo
