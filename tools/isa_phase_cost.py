#!/usr/bin/env python3
"""Static cost model per profiled phase of a kernel built with -DERL_PROFILE: the instruction stream between consecutive
s_memtime stamps, counted by class and priced with the measured single-wave costs (tools/mfma_issue_bench.hip):
v_mfma_f32_32x32x2 64 cycles, 16x16x4 32, plain VALU / accvgpr move / s_nop 4, transcendental 8; LDS / VMEM / SALU 0 (they
overlap).  Straight-line estimate: loops are counted once (pass --loop-mult N to scale segments that contain a backward
branch).  Usage: isa_phase_cost.py file.s kernel-substring"""
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and ":" in l)
    segs, cur = [], dict(M32=0, M16=0, v=0, t=0, a=0, n=0, r=0, w=0, g=0, s=0, br=0)
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        if op == "s_memtime":
            segs.append(cur)
            cur = dict.fromkeys(cur, 0)
        elif op.startswith("v_mfma_f32_32x32"): cur["M32"] += 1
        elif op.startswith("v_mfma"): cur["M16"] += 1
        elif op.startswith("v_accvgpr"): cur["a"] += 1
        elif op in ("v_exp_f32_e32", "v_rcp_f32_e32", "v_log_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32", "v_exp_f32", "v_rcp_f32"): cur["t"] += 1
        elif op.startswith("v_"): cur["v"] += 1
        elif op.startswith("s_nop"): cur["n"] += 1
        elif op.startswith("ds_read") or op.startswith("ds_bpermute"): cur["r"] += 1
        elif op.startswith("ds_"): cur["w"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"): cur["g"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): cur["br"] += 1
        elif op.startswith("s_"): cur["s"] += 1
    segs.append(cur)
    print(f"{'seg':>4} {'M32':>5} {'M16':>4} {'valu':>5} {'trans':>5} {'acc':>4} {'nop':>4} {'lds_r':>5} {'lds_w':>5} {'vmem':>4} {'salu':>5} {'br':>3} {'est.cycles':>10}")
    for i, c in enumerate(segs):
        est = 64 * c["M32"] + 32 * c["M16"] + 4 * (c["v"] + c["a"] + c["n"]) + 8 * c["t"]
        print(f"{i:4d} {c['M32']:5d} {c['M16']:4d} {c['v']:5d} {c['t']:5d} {c['a']:4d} {c['n']:4d} {c['r']:5d} {c['w']:5d} {c['g']:4d} {c['s']:5d} {c['br']:3d} {est:10d}")


if __name__ == "__main__":
    main()
