#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (--output-format csv): per kernel name, calls and mean / total of each counter.

    python tools/pmc_summary.py gpurun_out/pmc1 > profiles/r01_pmc_fetch.txt
"""
import csv, glob, os, re, sys
from collections import defaultdict

def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); return n[:90]

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not files:
    print("no *counter_collection.csv under", d); sys.exit(1)
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row.get("Kernel_Name", "?")); c = row.get("Counter_Name", "?"); v = float(row.get("Counter_Value", 0) or 0)
            a = acc[k][c]; a[0] += 1; a[1] += v
print("# %s" % ", ".join(os.path.relpath(f, d) for f in files))
print("%-92s %-14s %8s %16s %16s" % ("kernel", "counter", "calls", "mean", "total"))
rows = []
for k, cs in acc.items():
    for c, (n, t) in cs.items():
        rows.append((t, k, c, n))
for t, k, c, n in sorted(rows, reverse=True)[:60]:
    print("%-92s %-14s %8d %16.1f %16.1f" % (k, c, n, t / n, t))
