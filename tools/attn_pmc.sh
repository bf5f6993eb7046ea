#!/bin/bash
# SQ counters of the bf16 attention kernels on the three hot-path shapes (separate --pmc passes, nothing else traced)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/attn_pmc
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/attn_pmc/$tag -- python3 $R/tools/attn_bench.py > $R/gpurun_out/attn_pmc.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/gpurun_out/attn_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'attn' not in k: continue
        k = k[k.index('attn'):k.index('(')] + ' grid%sx%s' % (r['Grid_Size'], r.get('Workgroup_Size',''))
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    c = {n: sum(v)/len(v) for n, v in acc[k].items()}
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(k)
    print('   frac of wave cycles: ACTIVE_ANY %.2f (VALU %.2f LDS %.2f VMEM %.3f) WAIT_INST_ANY %.2f (LDS %.2f) WAIT_ANY %.2f' % tuple(c.get(n,0)/wc for n in ('SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_VMEM','SQ_WAIT_INST_ANY','SQ_WAIT_INST_LDS','SQ_WAIT_ANY')))
    g = c.get('GRBM_GUI_ACTIVE',1)/8
    print('   GUI cycles %.3g  MFMA busy/SIMD %.3f  LDS active/CU %.3f bank-conflict frac %.3f  waves %.0f  insts VALU %.3g (trans %.3g) SALU %.3g LDS %.3g VMEM %.3g' % (
        g, c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(g*1024), c.get('SQ_LDS_IDX_ACTIVE',0)/(g*256), c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1), c.get('SQ_WAVES',0),
        c.get('SQ_INSTS_VALU',0), c.get('SQ_INSTS_VALU_TRANS',0), c.get('SQ_INSTS_SALU',0), c.get('SQ_INSTS_LDS',0), c.get('SQ_INSTS_VMEM',0)))
PY
rm -rf $R/gpurun_out/attn_pmc
