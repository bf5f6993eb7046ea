cd tools/gemm_lab
for i in 1 2; do ./lab --sch=1 --dbg=0,4,1024 --quick --forms=1 enc_fc1 bert_inter dec_fc1 r3k768 2>&1 | grep "Q8" | cut -c1-110; done
