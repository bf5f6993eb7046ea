#!/usr/bin/env python3
"""Step time as a function of the CUs the weight-gradient persistent GEMM leaves free (ecamp_set_option
"p8_wgrad_reserve_cus"): the wgrad GEMMs run on a second stream beside the data-gradient chain, so the reserve is a static
split of the chip between the two.  Usage: python tools/reserve_probe.py 0 16 64 128"""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in [int(a) for a in sys.argv[1:]] or [0, 16, 64]:
    code = ("import sys; sys.argv=['bench.py','--steps','12','--warmup','3','--no-cpu-baseline','--no-prof'];"
            "sys.path.insert(0,%r); from ecamp_amd import hip_ops; hip_ops.set_option('p8_wgrad_reserve_cus',%d);"
            "import runpy; runpy.run_path(%r, run_name='__main__')" % (root, r, os.path.join(root, "bench.py")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip().splitlines()
    line = [l for l in out if l.startswith("{")]
    d = json.loads(line[-1]) if line else {}
    print("reserve %3d CUs: %.3f ms/step  %.1f pairs/s" % (r, d.get("ms_per_step", float("nan")), d.get("value", float("nan"))), flush=True)
