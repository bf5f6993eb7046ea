cd tools/gemm_lab
./lab --sch=1 --quick --forms=3 enc_fc1 ragged 2>&1 | grep -v "128^2" | cut -c1-130
