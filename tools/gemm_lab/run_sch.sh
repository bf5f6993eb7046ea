cd tools/gemm_lab
for i in 1 2; do ./lab --sch=1,2,3 --quick --forms=5 enc_fc1 enc_qkv bert_inter dec_fc1 dec_fc2 bert_dense enc_fc2 2>&1 | grep -v "128^2" | cut -c1-150; done
