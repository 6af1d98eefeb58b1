#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats: the kernel_stats.csv under a directory, printed with short kernel names (top 14 by time)."""
import csv
import glob
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]{0,60}>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
    if not files:
        print("no kernel_stats.csv under", sys.argv[1])
        return
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(f"{short(r['Name']):72s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:9.2f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms  {float(r['Percentage']):5.1f} %")


if __name__ == "__main__":
    main()
