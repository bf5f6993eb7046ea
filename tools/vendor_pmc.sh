#!/bin/bash
# The SAME counter passes over the vendor library's kernel (torch -> hipBLASLt; a yardstick only, never linked by the product) and
# over the persistent Q8 kernel, one shape per process so that every kernel name maps to one shape:
#   gpurun -- 'bash tools/vendor_pmc.sh'   ->  gpurun_out/vendor_pmc/<shape>/<pass>/**.csv, table in gpurun_out/r04_vendor_vs_q8_pmc.txt
# Separate --pmc passes, nothing else traced beside them; the timing pass is --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vendor_pmc
rm -rf $O; mkdir -p $O
SHAPES=("enc qkv" "enc fc1" "bert inter" "vocab")
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL")
python3 $R/tools/vendor_vs_q8.py "enc fc1" --n 3 > /dev/null 2>&1   # page the image in
for s in "${SHAPES[@]}"; do
  d=$O/$(echo $s | tr ' ' '_')
  mkdir -p $d
  python3 $R/tools/vendor_vs_q8.py "$s" --n 20 > $d/events.txt 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $d/trace -- python3 $R/tools/vendor_vs_q8.py "$s" > $d/trace.log 2>&1
  i=0
  for grp in "${GROUPS_[@]}"; do
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $d/pmc$i -- python3 $R/tools/vendor_vs_q8.py "$s" --n 2 > $d/pmc$i.log 2>&1 || echo "pass $i ($grp) failed on $s" >> $O/failed.txt
    i=$((i+1))
  done
done
python3 $R/tools/vendor_pmc_table.py $O > $R/gpurun_out/r04_vendor_vs_q8_pmc.txt 2>&1
# keep the merge-back small: drop the raw csv files, keep logs of failed passes
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
tail -n 80 $R/gpurun_out/r04_vendor_vs_q8_pmc.txt
