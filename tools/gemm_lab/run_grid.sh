cd tools/gemm_lab
for g in 0 200 150; do echo "grid $g"; ./lab --sch=1 --quick --forms=1 --grid=$g enc_fc1 2>&1 | grep "Q8" | cut -c1-110; done
for g in 0 225 150; do echo "grid $g"; ./lab --sch=1 --quick --forms=1 --grid=$g enc_qkv 2>&1 | grep "Q8" | cut -c1-110; done
for g in 0 226 198; do echo "grid $g"; ./lab --sch=1 --quick --forms=1 --grid=$g dec_fc1 2>&1 | grep "Q8" | cut -c1-110; done
for g in 0 192 128; do echo "grid $g"; ./lab --sch=1 --quick --forms=1 --grid=$g bert_dense 2>&1 | grep "Q8" | cut -c1-110; done
for g in 0 200 150; do echo "grid $g dgrad"; ./lab --sch=2 --quick --forms=2 --grid=$g enc_fc2 2>&1 | grep "Q8" | cut -c1-110; done
