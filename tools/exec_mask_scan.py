#!/usr/bin/env python3
"""Scan hipcc's assembly of a kernel for register saves executed under a reduced EXEC mask: a live-range split (v_accvgpr_write) or a
spill (scratch_store) placed between s_and_saveexec and the s_or_b64 that restores the mask saves only the lanes active there -- read back
under the full mask the other lanes hold garbage.  Found in round 4: the thread index of the (256, h2) PPO minibatch kernel was parked in
an AGPR at the end of a guarded LDS store whose mask was EMPTY for whole waves (the logged sums of one shape class were summed over stale
LDS).  The kernels avoid partial-mask blocks where they can; this scan is run over the assembly of every K6 translation unit after a
change:   hipcc ... --cuda-device-only -S -o k.s file.hip && python tools/exec_mask_scan.py k.s
(depth is reset at every s_barrier: barriers sit at the kernels' top level, full mask)"""
import re
import sys


def scan(path):
    depth, bad, out = 0, 0, []
    lines = open(path).read().split("\n")
    for i, ln in enumerate(lines):
        t = ln.strip()
        if t.startswith("s_barrier") or t.startswith("s_endpgm"):
            depth = 0
        elif t.startswith("s_and_saveexec"):
            depth += 1
        elif t.startswith("s_or_saveexec"):
            # the ELSE entry of an if / else lowered as s_and_saveexec ... s_or_saveexec ... s_or_b64 exec: still inside the SAME reduced-mask
            # region (counting it as a second opener left the depth stuck at 1 behind every if / else: false positives for the rest of
            # the kernel); an else without a preceding if does open one
            depth = max(depth, 1)
        elif t.startswith("s_or_b64 exec, exec,"):
            depth = max(0, depth - 1)
        elif depth > 0 and (t.startswith("v_accvgpr_write") or t.startswith("scratch_store")):
            m = re.match(r"v_accvgpr_write_b32 (a\d+), (v\d+)", t)
            # a value DEFINED inside the guarded block is a temporary of the active lanes; what matters is a save of a register last
            # written before the mask was reduced
            src, defined_inside = (m.group(2) if m else None), False
            if src:
                d = depth
                for j in range(i - 1, max(0, i - 400), -1):
                    tj = lines[j].strip()
                    if tj.startswith("s_and_saveexec"):
                        d -= 1
                        if d <= 0:
                            break
                    if re.match(r"^[a-z_0-9]+\s+" + src + r"\b", tj):
                        defined_inside = True
                        break
            if not defined_inside:
                bad += 1
                out.append(f"{path}:{i + 1}: {t}")
    return bad, out


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        bad, out = scan(p)
        total += bad
        for o in out[:20]:
            print(o)
        print(f"{p}: {bad} register saves of outer values under a reduced EXEC mask")
    sys.exit(1 if total else 0)
