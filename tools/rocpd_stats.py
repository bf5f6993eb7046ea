#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel totals, and GEMM launches grouped by grid shape.

    python tools/rocpd_stats.py gpurun_out/prof1/r1_results.db [--skip-first-frac 0.25] > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main():
    db = sys.argv[1]
    skip = float(sys.argv[sys.argv.index("--skip-first-frac") + 1]) if "--skip-first-frac" in sys.argv else 0.0
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, accum_vgpr_count from kernels order by start").fetchall()
    if not rows:
        print("no kernel dispatches")
        return
    t0, t1 = rows[0][1], rows[-1][2]
    cut = t0 + skip * (t1 - t0)
    rows = [r for r in rows if r[1] >= cut]
    tot = defaultdict(lambda: [0, 0.0])
    gem = defaultdict(lambda: [0, 0.0])
    for name, s, e, gx, gy, gz, wx, lds, vg, ag in rows:
        k = short(name)
        tot[k][0] += 1
        tot[k][1] += (e - s) / 1e3
        if "gemm_" in k:
            g = (k, gx // max(wx, 1), gz)
            gem[g][0] += 1
            gem[g][1] += (e - s) / 1e3
    wall = (rows[-1][2] - rows[0][1]) / 1e3
    busy = sum(v[1] for v in tot.values())
    # time with at least one kernel resident (union of the dispatch intervals) and the gaps between them
    union, gaps, cur_e = 0.0, [], None
    for r in rows:
        s_, e_ = r[1], r[2]
        if cur_e is None:
            cur_s, cur_e = s_, e_
        elif s_ <= cur_e:
            cur_e = max(cur_e, e_)
        else:
            union += cur_e - cur_s
            gaps.append(s_ - cur_e)
            cur_s, cur_e = s_, e_
    union += cur_e - cur_s
    gaps.sort()
    print("# %s  (dispatches after the first %.0f%% of the trace)" % (db, 100 * skip))
    print("# wall %.1f ms, sum of kernel durations %.1f ms, %d dispatches" % (wall / 1e3, busy / 1e3, len(rows)))
    if gaps:
        print("# device busy (union of dispatch intervals) %.1f ms = %.1f%% of wall; %d idle gaps: total %.1f ms, median %.2f us, p90 %.2f us, max %.1f us"
              % (union / 1e6, 100 * union / 1e3 / wall, len(gaps), sum(gaps) / 1e6, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))
    print("%-112s %7s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("%-112s %7d %12.1f %10.2f %6.2f" % (k, n, us, us / n, 100 * us / busy))
    print("\n# GEMM launches by (kernel, tiles = grid.x, split-K = grid.z)")
    for (k, tiles, gz), (n, us) in sorted(gem.items(), key=lambda kv: -kv[1][1])[:60]:
        print("%-60s tiles %6d splitk %3d calls %5d total_us %10.1f avg_us %9.2f" % (k[:60], tiles, gz, n, us, us / n))


if __name__ == "__main__":
    main()
