#!/bin/bash
# A/B of two library builds on the full step: tools/ab_bench.sh build/lib_old.so   (in-tree build = new)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-prof 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  ms %.2f  resident %.2f  fwd %.2f  fwd+bwd %.2f' % (d['value'], d['ms_per_step'], d['resident_ms_per_step'], d['fwd_only_ms'], d['fwd_bwd_ms']))"; }
echo -n "new: "; run
cp $R/ecamp_amd/libecamp_hip.so /tmp/lib_keep.so; cp $1 $R/ecamp_amd/libecamp_hip.so
echo -n "old: "; run
cp /tmp/lib_keep.so $R/ecamp_amd/libecamp_hip.so
echo -n "new: "; run
cp $1 $R/ecamp_amd/libecamp_hip.so
echo -n "old: "; run
cp /tmp/lib_keep.so $R/ecamp_amd/libecamp_hip.so
