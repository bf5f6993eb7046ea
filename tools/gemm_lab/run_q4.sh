# the runs behind profiles/r04_q4_and_power.txt sections 1, 3, 4 and 8 (one gpurun call each; from the repository root):
#   gpurun -- 'bash tools/gemm_lab/run_q4.sh sweep'    eight-wave (sch=1) against four-wave (sch=4, EARLY sch=5) kernels, K sweep + model shapes
#   gpurun -- 'bash tools/gemm_lab/run_q4.sh zero'     random against all-zero operands
#   gpurun -- 'bash tools/gemm_lab/run_q4.sh split'    decomposition of the four-wave loop (--dbg) and of the eight-wave epilogue (--dbg=4,1024)
cd tools/gemm_lab
case "${1:-sweep}" in
  sweep) ./lab --sch=4 --quick --forms=1 ragged 2>&1 | cut -c1-150 | tail -2
         for i in 1 2; do ./lab --sch=1,4,5 --quick --forms=1 ks256 ks512 ks1024 ks2048 sq4k r3k768 bert_inter enc_fc1 dec_fc1 2>&1 | grep "plain" | cut -c1-130; done ;;
  zero)  for i in 1 2; do
           ./lab --sch=1,4 --quick --forms=1 ks2048 sq4k 2>&1 | grep "plain" | cut -c1-110
           ./lab --zero --sch=1,4 --quick --forms=1 ks2048 sq4k 2>&1 | grep "plain" | sed 's/^/ZERO /' | cut -c1-110
         done ;;
  split) for i in 1 2; do
           ./lab --sch=4 --dbg=0,1,2,4,6,5 --quick --forms=1 ks2048 sq4k bert_inter 2>&1 | grep "plain" | cut -c1-110
           ./lab --sch=1 --dbg=0,4,1024 --quick --forms=1 enc_fc1 bert_inter dec_fc1 r3k768 2>&1 | grep "Q8" | cut -c1-110
         done ;;
esac
