#!/bin/bash
# A/B of an environment flag on the full step: tools/flag_ab.sh NAME=VALUE [NAME2=VALUE2 ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/bench.py --steps ${AB_STEPS:-12} --warmup 3 --no-cpu-baseline --no-prof 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  ms %.2f  resident %.2f  fwd %.2f  fwd+bwd %.2f' % (d['value'], d['ms_per_step'], d['resident_ms_per_step'], d['fwd_only_ms'], d['fwd_bwd_ms']))"; }
for i in $(seq 1 ${AB_ROUNDS:-2}); do
  echo -n "base    : "; run
  echo -n "flagged : "; env "$@" bash -c "$(declare -f run); R=$R; run"
done
