cd tools/gemm_lab
./lab --sch=5 --quick --forms=1 ragged 2>&1 | cut -c1-170 | tail -2
for i in 1 2; do ./lab --sch=1,4,5 --quick --forms=1 ks256 ks512 ks1024 ks2048 sq4k r3k768 bert_inter enc_fc1 dec_fc1 2>&1 | grep "plain" | cut -c1-130; done
