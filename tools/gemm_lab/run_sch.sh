cd tools/gemm_lab
./lab --sch=0,1,2 --quick --forms=2 enc_fc1 enc_qkv enc_fc2 bert_inter dec_fc1 dec_fc2 sq4k vocab 2>&1 | grep -v "128^2" | cut -c1-150
./lab --sch=0,1,2 --quick --forms=2 enc_fc1 enc_qkv enc_fc2 bert_inter dec_fc1 dec_fc2 sq4k vocab 2>&1 | grep -v "128^2" | cut -c1-150
