#!/usr/bin/env python3
"""Folds one box's tools/r05_wg_map.sh output (gpurun_out/r05_wgmap_<tag>/) into a compact record for profiles/:
    python tools/wgmap_summarise.py gpurun_out/r05_wgmap_p5 > profiles/r05_wgmap_slow_box.json"""
import glob
import json
import os
import sys

d = sys.argv[1]
out = {"source": os.path.basename(d.rstrip("/")), "script": "tools/r05_wg_map.sh (tools/clock_probe.hip, tools/k6_wg_map_ab.py, bench.py --steps 20 --warmup 5 --repeats 3)"}
try:
    cp = json.load(open(os.path.join(d, "clock_probe.json")))
    out["clock_probe"] = {k: cp.get(k) for k in ("code_walk_56KB", "k6_like_forward_mix", "mfma_bf16_32x32x16", "valu_fma_f32", "lds_read_b128")}
except Exception as e:                                      # boxes sampled before the probe had the code walk
    out["clock_probe"] = None
ab = [json.loads(ln) for ln in open(os.path.join(d, "ab.jsonl"))]
out["k6_in_update_loop"] = [{k: r.get(k) for k in ("wg_map", "sampled", "loop_ms_40_minibatches", "us_span", "workgroup_us", "shader_mhz", "dur_us", "dur_us_mean_by_xcc",
                                                    "params_checksum") if k in r} for r in ab]
out["bench_c4"] = {}
for f in sorted(glob.glob(os.path.join(d, "c4_*.json"))):
    b = json.loads(open(f).readline())
    r, br = b["roofline"], b["breakdown"]
    out["bench_c4"][os.path.basename(f)[:-5]] = {
        "value": b["value"], "ms_per_step": b["ms_per_step"], "repeated_regions_ms_per_step": b["extra"]["repeated_regions_ms_per_step"],
        "workgroup_map": r.get("workgroup_map"), "k6_sampled_launch_us": r["avg_launch_us"], "k6_us_upper_bound_by_difference": br.get("k6_us_upper_bound_by_difference"),
        "shader_mhz": r["shader_mhz"], "slab_reduce_us": br["slab_reduce_us"], "clip_adam_us": br["clip_adam_us"], "explore_env_ms": br["explore_env_ms"],
        "update_net_ms": br["update_net_ms"], "phase_cycles": r["phase_cycles"]}
print(json.dumps(out, indent=1))
