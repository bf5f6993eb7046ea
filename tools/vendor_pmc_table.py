#!/usr/bin/env python3
"""One table per shape from tools/vendor_pmc.sh's output: the GEMM kernels that ran (vendor library and Q8, forward and data-gradient
forms), their duration and kernel-descriptor resources from the trace pass, every counter per launch and per (256 x 256 x 64) tile step.

    python3 tools/vendor_pmc_table.py gpurun_out/vendor_pmc > profiles/r04_vendor_vs_q8_pmc.txt
"""
import csv, glob, os, re, sys
from collections import defaultdict

SHAPES = {"enc_qkv": (12800, 2304, 768), "enc_fc1": (12800, 3072, 768), "bert_inter": (32768, 1536, 768), "vocab": (32768, 30000, 768)}
csv.field_size_limit(1 << 30)


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n


def is_gemm(n):
    return n.startswith("gemm_bf16") or n.startswith("Cijk_") or "Cijk" in n


root = sys.argv[1]
for shape in sorted(os.listdir(root)):
    d = os.path.join(root, shape)
    if not os.path.isdir(d) or shape not in SHAPES:
        continue
    M, N, K = SHAPES[shape]
    steps = ((M + 255) // 256) * ((N + 255) // 256) * (K // 64)
    print("=" * 150)
    print("shape %s  M=%d N=%d K=%d   256^2 tiles %d, (tile x 64-deep K tile) steps %d, %.2f rounds on 256 CUs" %
          (shape, M, N, K, steps // (K // 64), steps, steps / (K // 64) / 256.0))
    ev = os.path.join(d, "events.txt")
    if os.path.exists(ev):
        print(open(ev).read().rstrip())
    # trace pass: per kernel name -> durations (ordered), resources
    kern = {}
    order = []
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                n = short(r["Kernel_Name"])
                if not is_gemm(n):
                    continue
                k = kern.setdefault(n, {"dur": [], "res": None})
                if n not in order:
                    order.append(n)
                k["dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                k["res"] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                            r["Workgroup_Size_X"], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
    # counters: per kernel name -> counter -> mean
    cnt = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                n = short(r["Kernel_Name"])
                if not is_gemm(n):
                    continue
                if n not in order:
                    order.append(n)
                a = cnt[n][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"] or 0)
    names = order
    print()
    for i, n in enumerate(names):
        k = kern.get(n)
        if k:
            ds = sorted(k["dur"])
            print("[%d] %s" % (i, n[:400]))
            print("     launches %d  median %.1f us  min %.1f us   VGPR %s  AGPR %s  SGPR %s  LDS %s B  scratch %s  workgroup %s  grid %d threads = %d workgroups"
                  % (len(ds), ds[len(ds) // 2], ds[0], *k["res"][:6], k["res"][6], k["res"][6] // max(1, int(k["res"][5]))))
        else:
            print("[%d] %s   (not in the trace pass)" % (i, n[:400]))
    counters = sorted({c for n in names for c in cnt[n]})
    print()
    print("%-30s" % "counter (mean per launch)" + "".join("%18s" % ("[%d]" % i) for i in range(len(names))) + "   | per tile step:" + "".join("%12s" % ("[%d]" % i) for i in range(len(names))))
    for c in counters:
        vals = [cnt[n][c][1] / cnt[n][c][0] if cnt[n][c][0] else float("nan") for n in names]
        print("%-30s" % c + "".join("%18.4g" % v for v in vals) + "   |               " + "".join("%12.4g" % (v / steps) for v in vals))
    # derived
    def g(n, c):
        a = cnt[n].get(c)
        return a[1] / a[0] if a and a[0] else float("nan")
    print()
    print("derived:")
    for i, n in enumerate(names):
        wc, busy, mf = g(n, "SQ_WAVE_CYCLES"), g(n, "SQ_BUSY_CYCLES"), g(n, "SQ_VALU_MFMA_BUSY_CYCLES")
        gui = g(n, "GRBM_GUI_ACTIVE")
        print("  [%d] MFMA busy / (4 SIMD x 256 CU x GUI cycles per XCD) = %.3f   WAIT_ANY/WAVE_CYCLES = %.3f   WAIT_INST_ANY/WAVE_CYCLES = %.3f   ACTIVE_INST_ANY/WAVE_CYCLES = %.3f"
              "   fetch MB = %.1f (x2 gfx950 correction: %.1f)   write MB = %.1f   L2 hit = %.3f" %
              (i, mf / (1024.0 * gui / 8.0) if gui == gui else float("nan"), g(n, "SQ_WAIT_ANY") / wc, g(n, "SQ_WAIT_INST_ANY") / wc, g(n, "SQ_ACTIVE_INST_ANY") / wc,
               g(n, "FETCH_SIZE") / 1024.0, g(n, "FETCH_SIZE") / 512.0, g(n, "WRITE_SIZE") / 1024.0,
               g(n, "TCC_HIT_sum") / max(1.0, g(n, "TCC_HIT_sum") + g(n, "TCC_MISS_sum"))))
    print()
