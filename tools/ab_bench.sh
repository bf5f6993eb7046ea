#!/bin/bash
# A/B of two library builds on the full step: tools/ab_bench.sh build/lib_old.so   (in-tree build = new; the old one through ECAMP_LIB)
R=${GRAFT_REPO_ROOT:-/root/repo}
OLD=$(readlink -f $1)
run() { python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-prof 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  ms %.2f  resident %.2f  fwd %.2f  fwd+bwd %.2f' % (d['value'], d['ms_per_step'], d['resident_ms_per_step'], d['fwd_only_ms'], d['fwd_bwd_ms']))"; }
for i in 1 2; do
  echo -n "new: "; run
  echo -n "old: "; ECAMP_LIB=$OLD run
done
