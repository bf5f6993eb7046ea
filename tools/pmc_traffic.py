#!/usr/bin/env python3
"""HBM-side traffic of the GEMM family per launch, from two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench step.

    gpurun -- 'bash tools/pmc_traffic.sh'          # runs the passes on the GPU box (nothing else traced), then this script
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r03_pmc_traffic.json

Correction per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> doubled; WRITE_SIZE as
reported; both counters are in KB."""
import csv, glob, json, os, re, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecamp_amd.build import gemm_source_hash


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = re.sub(r"^void ", "", row["Kernel_Name"])
            k = re.sub(r"\(.*$", "", k)
            a = acc[k]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
per, n, fb, wb = {}, 0, 0.0, 0.0
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("gemm_bf16"):
        continue
    c = max(fetch.get(k, [0])[0], write.get(k, [0])[0])
    f = fetch.get(k, [0, 0.0])[1] * 1024 * 2
    w = write.get(k, [0, 0.0])[1] * 1024
    per[k] = {"calls": c, "fetch_bytes_per_launch": f / max(c, 1), "write_bytes_per_launch": w / max(c, 1)}
    n += c; fb += f; wb += w
print(json.dumps({"gemm_source_sha256": gemm_source_hash(),
                  "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof (ECAMP_OVERLAP_WGRAD=0 ECAMP_OVERLAP_BRANCHES=0), separate passes",
                  "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; both in KB",
                  "gemm_launches": n, "gemm_fetch_bytes_per_launch": fb / max(n, 1), "gemm_write_bytes_per_launch": wb / max(n, 1),
                  "per_kernel": per, "gemm_traffic_bytes_per_launch": (fb + wb) / max(n, 1)}, indent=1))
