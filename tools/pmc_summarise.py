#!/usr/bin/env python3
"""Summarise rocprofv3 counter-collection CSVs (one --pmc pass per file) into per-kernel, per-dispatch averages.
   python tools/pmc_summarise.py out.json pass1/p_counter_collection.csv pass2/p_counter_collection.csv ...
Rows of one (dispatch, counter) -- one per XCD / dimension instance -- are summed, then averaged over the dispatches of
the kernel.  FETCH_SIZE / WRITE_SIZE (KB) are turned into HBM bytes with the gfx950 correction of MI355X_MICROARCH.md
(FETCH_SIZE counts 128-byte read requests as 64 bytes: x2; WRITE_SIZE as reported)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import KERNEL_SOURCES, kernel_source_sha16  # noqa: E402  (stamp = hash of the sources the counters were collected on)


def short(name: str) -> str:
    # PMC_KEEP_TEMPLATE=1 keeps a kernel's template arguments (critic_tile_kernel<1, 2, 2>: the three passes of the SAC critic apart)
    if os.environ.get("PMC_KEEP_TEMPLATE"):
        m = re.search(r"(\w+_kernel(?:<[^(]*>)?)", name)
        return m.group(1).replace(" ", "") if m else name[:60]
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:60]


def case_names():
    """PMC_CASES=<kernel>:<launches per case>:<name0>,<name1>,...: the kernel's dispatches, in order, belong to consecutive cases of
    that many launches each (a workload that runs one kernel at several sizes: tools/c3_pmc_workload.py C3_PART=k9)"""
    spec = os.environ.get("PMC_CASES")
    if not spec:
        return None
    k, per, names = spec.split(":")
    return k, int(per), names.split(",")


def main():
    out_path, files = sys.argv[1], sys.argv[2:]
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> value
    dur = defaultdict(dict)
    cases = case_names()
    for f in files:
        with open(f, newline="") as fh:
            rows = list(csv.DictReader(fh))
        order = {}
        if cases:       # dispatch ids of the case kernel in this file, in launch order -> case index
            ids = sorted({int(r["Dispatch_Id"]) for r in rows if short(r["Kernel_Name"]) == cases[0]})
            order = {d: i // cases[1] for i, d in enumerate(ids)}
        if True:
            for row in rows:
                k = short(row["Kernel_Name"])
                if cases and k == cases[0]:
                    ci = order[int(row["Dispatch_Id"])]
                    k = f"{k}[{cases[2][ci] if ci < len(cases[2]) else ci}]"
                did = (f, row["Dispatch_Id"])
                per[k][row["Counter_Name"]][did] += float(row["Counter_Value"])
                if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    dur[k][did] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    res = {}
    for k, counters in per.items():
        e = {c: round(sum(v.values()) / len(v), 1) for c, v in sorted(counters.items())}
        e["dispatches"] = max(len(v) for v in counters.values())
        if dur[k]:
            e["avg_duration_us"] = round(sum(dur[k].values()) / len(dur[k]), 2)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_read_bytes"] = int(e["FETCH_SIZE"] * 1024 * 2)
            e["hbm_write_bytes"] = int(e["WRITE_SIZE"] * 1024)
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "avg_duration_us" in e:
            # the counter sums busy cycles over all SIMDs (256 CUs x 4); gfx950 engine clock 2.4 GHz
            e["mfma_busy_cycles_per_simd"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024, 1)
            e["mfma_busy_frac_of_kernel_time_at_2.4GHz"] = round(e["mfma_busy_cycles_per_simd"] / (e["avg_duration_us"] * 2400), 3)
        if k in KERNEL_SOURCES:
            e["source_sha16"] = kernel_source_sha16(k)
        res[k] = e
    json.dump({"note": "rocprofv3 --kernel-trace --pmc <one counter set per pass>; per-dispatch averages, summed over XCDs "
                       "(tools/pmc_summarise.py)", "kernels": res}, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
