for d in 0 1 2 4 8 16 32 64 128 127 255; do echo -n "dbg $d: "; ECAMP_SR_DBG=$d python tools/sr_bench.py 2>&1 | grep "mode 1" ; done
