#!/usr/bin/env python3
"""Kernel statistics (calls, total / average / min / max duration) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats` writes `<name>_results.db` on ROCm 7.2).  Output: CSV on stdout, the same columns as
rocprofv3's `kernel_stats.csv`.  Optionally split by grid size (--by-grid) so that one kernel name at several problem
sizes does not get averaged into one line."""
import argparse
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--by-grid", action="store_true")
    ap.add_argument("--top", type=int, default=0)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    grid = ", grid_x || 'x' || grid_y || 'x' || grid_z" if a.by_grid else ""
    grp = f"{name}{', grid_x, grid_y, grid_z' if a.by_grid else ''}"
    q = (f"select {name}{grid}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels "
         f"group by {grp} order by sum(end - start) desc")
    rows = list(c.execute(q))
    total = sum(r[-4] for r in rows) or 1
    w = sys.stdout.write
    w('"Name",' + ('"Grid",' if a.by_grid else "") + '"Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for r in rows[: a.top or None]:
        nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
        nm = nm[: nm.find("(")] if "(" in nm else nm
        rest = r[1:] if not a.by_grid else r[2:]
        calls, tot, avg, mn, mx = rest
        w(f'"{nm}",' + (f'"{r[1]}",' if a.by_grid else "") + f"{calls},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx}\n")


if __name__ == "__main__":
    main()
