#!/usr/bin/env python3
"""Where is the GPU idle inside a PPO iteration?  Reads a rocprofv3 --kernel-trace CSV of `python bench.py ...` and prints, per
kernel name, the idle time BEFORE its launches (start - previous kernel's end), and the iteration's busy / idle split.
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep
    python tools/iteration_gaps.py out/.../t_kernel_trace.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: (re.search(r"(\w+_kernel|\w+Buffer\w*)", n) or re.search(r"(\w+)", n)).group(1)
gap, cnt, dur = defaultdict(float), defaultdict(int), defaultdict(float)
prev_end = None
# only the steady part: from the 10th rollout launch on
starts = [i for i, r in enumerate(rows) if "rollout_fused" in r["Kernel_Name"]]
lo, hi = starts[10], starts[-2]
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    if prev_end is not None:
        gap[n] += max(0, s - prev_end)
    cnt[n] += 1
    dur[n] += e - s
    prev_end = max(prev_end or 0, e)
iters = len([i for i in starts if lo <= i < hi])
wall = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
print(f"{iters} iterations, {wall / iters / 1e3:.1f} us each; busy {sum(dur.values()) / iters / 1e3:.1f} us, idle {sum(gap.values()) / iters / 1e3:.1f} us")
for n in sorted(gap, key=lambda k: -gap[k]):
    print(f"  {n:36s} launches/iter {cnt[n] / iters:6.1f}  busy {dur[n] / iters / 1e3:8.1f} us  idle before {gap[n] / iters / 1e3:7.1f} us  ({gap[n] / cnt[n] / 1e3:5.2f} us each)")
