#!/bin/bash
# interleaved A/B of one environment flag on the full step: tools/flag_ab2.sh NAME=VALUE  (base first, then flagged, AB_ROUNDS times)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline --no-prof 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  ms %.2f  resident %.2f  fwd %.2f  fwd+bwd %.2f' % (d['value'], d['ms_per_step'], d['resident_ms_per_step'], d['fwd_only_ms'], d['fwd_bwd_ms']))"; }
for i in $(seq 1 ${AB_ROUNDS:-3}); do
  echo -n "base    : "; run
  echo -n "$* : "; env "$@" python3 $R/bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline --no-prof 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  ms %.2f  resident %.2f  fwd %.2f  fwd+bwd %.2f' % (d['value'], d['ms_per_step'], d['resident_ms_per_step'], d['fwd_only_ms'], d['fwd_bwd_ms']))"
done
