R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or wgrad" 2>&1 | tail -3
echo "== wgrad group new"; python tools/wgrad_group_bench.py
cp ecamp_amd/libecamp_hip.so /tmp/new.so; cp build/lib_old.so ecamp_amd/libecamp_hip.so
echo "== wgrad group old"; python tools/wgrad_group_bench.py
cp /tmp/new.so ecamp_amd/libecamp_hip.so
echo "== MT auto"; python tools/gemm_bench.py "enc proj" "enc fc2" "bert dense" "bert out" "dec proj" "dec fc2"
echo "== MT=4"; ECAMP_Q8_MT=4 python tools/gemm_bench.py "enc proj" "enc fc2" "bert dense" "bert out" "dec proj" "dec fc2"
echo "== MT=3"; ECAMP_Q8_MT=3 python tools/gemm_bench.py "enc proj" "enc fc2" "bert dense" "bert out" "dec proj" "dec fc2"
bash tools/ab_bench.sh build/lib_old.so
