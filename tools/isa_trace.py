#!/usr/bin/env python3
"""Compressed view of a kernel's instruction stream from hipcc's -save-temps assembly: one letter per instruction class,
run-length encoded (M mfma, v valu, a accvgpr move, r ds_read, w ds_write, G global/buffer load, T global store, S scratch,
s salu, B barrier, W waitcnt, b branch).  Usage: isa_trace.py file.s kernel-name-substring [--wide]"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith("v_accvgpr"): return "a"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "r"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "w"
    if op.startswith("ds_"): return "d"
    if op.startswith("scratch_"): return "S"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): return "G"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"): return "T"
    if op.startswith("global_") or op.startswith("buffer_"): return "A"
    if op == "s_barrier": return "B"
    if op.startswith("s_waitcnt"): return "W"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "b"
    if op.startswith("s_nop"): return "n"
    if op.startswith("s_"): return "s"
    if op.startswith("v_"): return "v"
    return "?"


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and ":" in l)
    out, labels = [], {}
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".LBB"):
                out.append("|")
            continue
        out.append(classify(t.split()[0]))
    s = "".join(out)
    rle = re.sub(r"(.)\1+", lambda m: f"{m.group(1)}{len(m.group(0))}", s)
    counts = {c: s.count(c) for c in sorted(set(s))}
    print(counts)
    if "--wide" in sys.argv:
        for i in range(0, len(rle), 160):
            print(rle[i:i + 160])


if __name__ == "__main__":
    main()
