#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of the bench command -> profiles/rNN_kernel_times.json: per kernel the average duration, the
number of calls and the hash of the sources the kernel was built from (bench.py reports the number as `roofline.kernel_us_rocprof`
and drops it when the kernel's sources have changed since).
   python tools/kstats_summarise.py out.json path/to/x_kernel_stats.csv ["command that was profiled"]"""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import KERNEL_SOURCES, kernel_source_sha16  # noqa: E402


def main():
    out_path, csv_path = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --gpus 1 --steps 20 --warmup 5"
    res = {}
    for row in csv.DictReader(open(csv_path, newline="")):
        m = re.search(r"(\w+_kernel)", row["Name"])
        if not m:
            continue
        k = m.group(1)
        e = res.setdefault(k, {"calls": 0, "total_us": 0.0})
        e["calls"] += int(row["Calls"])
        e["total_us"] += float(row["TotalDurationNs"]) * 1e-3
    for k, e in res.items():
        e["avg_us"] = round(e["total_us"] / e["calls"], 3)
        e["total_us"] = round(e["total_us"], 1)
        if k in KERNEL_SOURCES:
            e["source_sha16"] = kernel_source_sha16(k)
    json.dump({"note": f"rocprofv3 --kernel-trace --stats -- {cmd} (tools/kstats_summarise.py; instantiations of one kernel template merged)",
               "csv": os.path.basename(csv_path), "kernels": res}, open(out_path, "w"), indent=1)
    for k in sorted(res, key=lambda k: -res[k]["total_us"])[:10]:
        print(f"{k:40s} calls {res[k]['calls']:6d} avg {res[k]['avg_us']:9.2f} us")


if __name__ == "__main__":
    main()
