# the PCIe-inclusive step against the resident one, several fresh processes (is `value` stable?)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4 5; do python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-prof 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  ms %.2f  resident %.2f' % (d['value'], d['ms_per_step'], d['resident_ms_per_step']))"; done
