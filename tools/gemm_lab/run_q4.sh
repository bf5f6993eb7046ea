cd tools/gemm_lab
for i in 1 2; do
./lab --sch=1,4 --quick --forms=1 ks2048 sq4k 2>&1 | grep "plain" | cut -c1-110
./lab --zero --sch=1,4 --quick --forms=1 ks2048 sq4k 2>&1 | grep "plain" | sed 's/^/ZERO /' | cut -c1-110
done
