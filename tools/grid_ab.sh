#!/bin/bash
# same-box A/B of the persistent GEMM's data-gradient grid: persistent (one workgroup per CU) against one item per workgroup (the dispatcher as the queue)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  ms %.2f  resident %.2f  fwd %.2f  fwd+bwd %.2f' % (d['value'], d['ms_per_step'], d['resident_ms_per_step'], d['fwd_only_ms'], d['fwd_bwd_ms']))"; }
for i in 1 2; do
  echo -n "persistent            : "; run
  echo -n "bwd one item per wg   : "; ECAMP_Q8_BWD_GRID=1000000 run
  echo -n "bwd grid 512          : "; ECAMP_Q8_BWD_GRID=512 run
done
