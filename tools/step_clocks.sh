#!/bin/bash
# Shader clock of every kernel of the bench step, in place: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration, from ONE
# rocprofv3 pass that collects the counter and the kernel trace together (PMC collection serializes the dispatches, so this is the
# serialized step; the production step interleaves two streams).   gpurun -- 'bash tools/step_clocks.sh'  ->  gpurun_out/r04_step_clocks.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/step_clocks
rm -rf $O
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof --only-value > $R/gpurun_out/step_clocks.log 2>&1
python3 - <<PY > $R/gpurun_out/r04_step_clocks.txt
import csv, glob, re, collections
cnt = {}
for f in glob.glob('$O/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            cnt[r['Dispatch_Id']] = (float(r['Counter_Value']), r['Kernel_Name'], r.get('Start_Timestamp'), r.get('End_Timestamp'))
dur = {}
for f in glob.glob('$O/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
ids = sorted(cnt, key=lambda k: int(k))
ids = ids[len(ids) // 3:]          # skip the warm-up steps
for k in ids:
    c, name, s, e = cnt[k]
    d = dur.get(k, (None,))[0]
    if d is None and s and e: d = int(e) - int(s)
    if not d: continue
    n = re.sub(r'^void ', '', name); n = re.sub(r'\(.*$', '', n)[:84]
    a = acc[n]; a[0] += 1; a[1] += c / 8.0; a[2] += d
print('# shader clock per kernel inside the (serialized, counter-collecting) bench step: sum(GRBM_GUI_ACTIVE / 8) / sum(duration)')
print('%-86s %6s %10s %9s' % ('kernel', 'calls', 'avg_us', 'GHz'))
for n, (k, c, d) in sorted(acc.items(), key=lambda kv: -kv[1][2])[:40]:
    print('%-86s %6d %10.1f %9.3f' % (n, k, d / k / 1e3, c / d))
tot_c = sum(v[1] for v in acc.values()); tot_d = sum(v[2] for v in acc.values())
print('%-86s %6s %10s %9.3f' % ('all kernels (time-weighted)', '', '', tot_c / tot_d))
PY
rm -rf $O
head -30 $R/gpurun_out/r04_step_clocks.txt
