#!/bin/bash
# A/B of the Q8 stream variants on the full step (ECAMP_Q8_SCH: bit 0 forward, 1 data-gradient, 2 weight-gradient, 3 grouped weight gradients)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline --no-prof 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  ms %.2f  resident %.2f  fwd %.2f  fwd+bwd %.2f' % (d['value'], d['ms_per_step'], d['resident_ms_per_step'], d['fwd_only_ms'], d['fwd_bwd_ms']))"; }
for i in 1 2; do
  for m in 0 5 7 15; do
    echo -n "ECAMP_Q8_SCH=$m : "; ECAMP_Q8_SCH=$m run
  done
done
