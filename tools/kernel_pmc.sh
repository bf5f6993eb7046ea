#!/bin/bash
# SQ counters of every kernel whose name contains $1, running `python3 $2 [args...]` (separate --pmc passes, nothing else traced)
#   tools/kernel_pmc.sh sr_ tools/sr_bench.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MATCH=$1; shift
SCRIPT=$1; shift
OUT=$R/gpurun_out/kpmc
rm -rf $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/kpmc.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if '$MATCH' not in k: continue
        k = k[:k.index('(')] if '(' in k else k
        k = k.replace('void ', '') + ' grid%sx%s' % (r['Grid_Size'], r.get('Workgroup_Size',''))
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    c = {n: sum(v)/len(v) for n, v in acc[k].items()}
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(k)
    print('   frac of wave cycles: ACTIVE_ANY %.2f (VALU %.2f LDS %.2f VMEM %.3f MISC %.3f) WAIT_INST_ANY %.2f (LDS %.2f) WAIT_ANY %.2f' % tuple(c.get(n,0)/wc for n in ('SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_VMEM','SQ_ACTIVE_INST_MISC','SQ_WAIT_INST_ANY','SQ_WAIT_INST_LDS','SQ_WAIT_ANY')))
    g = c.get('GRBM_GUI_ACTIVE',1)/8
    print('   GUI cycles %.3g  MFMA busy/SIMD %.3f  LDS active/CU %.3f bank-conflict frac %.3f unaligned %.3g addr-conflict %.3g  waves %.0f' % (
        g, c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(g*1024), c.get('SQ_LDS_IDX_ACTIVE',0)/(g*256), c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1),
        c.get('SQ_LDS_UNALIGNED_STALL',0), c.get('SQ_LDS_ADDR_CONFLICT',0), c.get('SQ_WAVES',0)))
    print('   insts: VALU %.4g (trans %.3g, MFMA %.4g) SALU %.4g LDS %.4g VMEM %.4g' % (
        c.get('SQ_INSTS_VALU',0), c.get('SQ_INSTS_VALU_TRANS',0), c.get('SQ_INSTS_MFMA',0), c.get('SQ_INSTS_SALU',0), c.get('SQ_INSTS_LDS',0), c.get('SQ_INSTS_VMEM',0)))
PY
rm -rf $OUT
