#!/usr/bin/env python3
"""One JSON record of THIS box: what it is (rocm-smi / amd-smi: clocks, power cap, partition modes), what clock it gives each kind of
work (tools/bin/clock_probe), and how the PPO minibatch kernel runs on it -- back to back and inside the config-4 loop: microseconds, the
shader clock inside the kernel, cycles per phase (the library's own stamps on sampled launches, erl_k6_timing_clocks).  Round 5: the
driver's boxes ran the kernel 15-40 % slower than the boxes it was tuned on (BENCH_r02..r04); every gpurun call of the round leaves one
of these, and a slow and a fast one are committed as profiles/r05_box_*.json.

    python tools/box_record.py > gpurun_out/box_<tag>.json
"""
import json
import os
import subprocess
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (smi_snapshot)
from elegantrl_amd import _hip, ops  # noqa: E402

dev = th.device("cuda:0")
out = {"time": time.strftime("%Y-%m-%d %H:%M:%S"), "device": th.cuda.get_device_name(0), "host": os.uname().nodename}


def sh(cmd, timeout=60):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=timeout).stdout.strip()
    except Exception as e:
        return f"error: {e!r}"


out["smi_idle"] = bench.smi_snapshot()
out["rocminfo_gpu"] = [ln.strip() for ln in sh("rocminfo").splitlines()
                       if any(k in ln for k in ("Marketing Name", "Uuid", "Compute Unit", "Max Clock", "Memory Properties", "Name:                    gfx"))][:16]
out["partition"] = {"compute": sh("rocm-smi --showcomputepartition | grep -i partition"), "memory": sh("rocm-smi --showmemorypartition | grep -i partition")}
probe = os.path.join(ROOT, "tools", "bin", "clock_probe")
if os.path.exists(probe):
    txt = sh(probe, 120)
    try:
        out["clock_probe"] = json.loads(txt.splitlines()[-1])
    except Exception:
        out["clock_probe"] = {"raw": txt[-400:]}

# ---- the minibatch kernel alone, back to back (config 4's shape), every launch sampled
N, S, A, H, B, h1, h2 = 4096, 64, 8, 32, 16384, 128, 128
g = th.Generator(device=dev).manual_seed(0)
sa, sc = ops.MlpSpec(S, h1, h2, A, True), ops.MlpSpec(S, h1, h2, 1, False)
Pa = sa.count
flat = th.randn(Pa + sc.count, device=dev, generator=g) * 0.05
avg, std = th.zeros(S, device=dev), th.ones(S, device=dev)
states = th.randn((H, N, S), device=dev, generator=g)
actions = th.randn((H, N, A), device=dev, generator=g)
logprobs = th.randn((H, N), device=dev, generator=g) - 8
adv, ret = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
um = th.rand((H, N), device=dev, generator=g) < 0.995
idsets = [th.randint(H * N, (B,), device=dev, generator=g) for _ in range(8)]
stride, n_slabs = ops.ppo_slab_stride(S, h1, h2, A), ops.ppo_num_slabs(B)
slabs = th.empty((n_slabs, stride), device=dev)


def k6(i):
    ops.ppo_step(flat[:Pa], flat[Pa:], avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret, idsets[i % 8], 0.25, 0.001,
                 1.0 / B, slabs, n_slabs)


res = {}
for arith in ("split", "f32"):
    ops.ppo_set_arith(arith)
    for i in range(300):
        k6(i)
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(400):
        k6(i)
    e1.record()
    th.cuda.synchronize()
    back_to_back = e0.elapsed_time(e1) * 1e3 / 400
    _hip.k6_timing_enable(4)          # every launch leaves its span; every 4th also sits inside an event bracket
    for i in range(400):
        k6(i)
    th.cuda.synchronize()
    _hip.k6_timing_enable(False)
    ev_s, span_s, n = _hip.k6_timing_read2()
    c, cb = _hip.k6_timing_clocks(False), _hip.k6_timing_clocks(True)
    res[arith] = {"us_back_to_back_events": round(back_to_back, 2), "us_span_unbracketed": round(c["span_us"] or 0, 2),
                  "us_span_bracketed": round(span_s / max(n, 1) * 1e6, 2),
                  "us_event_bracket": round(ev_s / max(n, 1) * 1e6, 2), "shader_mhz": round(c["shader_mhz"], 1),
                  "shader_mhz_bracketed": round(cb["shader_mhz"], 1),
                  "workgroup_us": round(c["workgroup_us"], 2), "workgroup_us_bracketed": round(cb["workgroup_us"], 2),
                  "phase_cycles": {k: round(v) for k, v in c["phase_cycles"].items()},
                  "phase_cycles_bracketed": {k: round(v) for k, v in cb["phase_cycles"].items()},
                  "workgroups": _hip.k6_wg_summary(_hip.k6_timing_last_records(False)),
                  "note": "stand-alone erl_ppo_step_f32 (every workgroup splits W1 / W2 itself: no images from the update loop)"}
# ---- the same kernel with the caches scrubbed in front of every launch (a 512 MiB copy: L2 and MALL hold nothing of the kernel's code,
# weights or rows) -- what the kernel pays for a cold start on this box
ops.ppo_set_arith("split")
scrub_a = th.empty(1 << 27, dtype=th.float32, device=dev)
scrub_b = th.empty_like(scrub_a)
_hip.k6_timing_enable(2)
for i in range(60):
    scrub_b.copy_(scrub_a)
    k6(i)
th.cuda.synchronize()
_hip.k6_timing_enable(False)
_hip.k6_timing_read2()
c = _hip.k6_timing_clocks(False)
res["split_cold_caches"] = {"us_span_unbracketed": round(c["span_us"] or 0, 2), "shader_mhz": round(c["shader_mhz"], 1), "workgroup_us": round(c["workgroup_us"], 2),
                            "phase_cycles": {k: round(v) for k, v in c["phase_cycles"].items()},
                            "workgroups": _hip.k6_wg_summary(_hip.k6_timing_last_records(False))}
del scrub_a, scrub_b
ops.ppo_set_arith("auto")
# ---- the kernel as the update loop launches it (erl_ppo_update_f32: weight images from the loop, slab reduction and clip + Adam between
# the launches, lr = 0), 3 x 40 minibatches on the same random rollout
m1, m2 = th.zeros_like(flat), th.zeros_like(flat)
rows = th.zeros((40, stride), device=dev)
ids40 = th.stack([idsets[i % 8] for i in range(40)])
for rep in range(3):
    if rep == 2:
        _hip.k6_timing_enable(4)
        _hip.kernel_span_enable(4)
    ops.ppo_update(flat, m1, m2, avg, std, avg, std, S, h1, h2, A, states, actions, um, logprobs, adv, ret, ids40, 0.25, 0.001, slabs, rows, 1 + 40 * rep,
                   0.0, 3.0)
th.cuda.synchronize()
_hip.k6_timing_enable(False)
ev_s, span_s, n = _hip.k6_timing_read2()
c = _hip.k6_timing_clocks(False)
red_us, _n1 = _hip.kernel_span_read(_hip.SPAN_SLAB_REDUCE)
adam_us, _n2 = _hip.kernel_span_read(_hip.SPAN_CLIP_ADAM)
_hip.kernel_span_enable(False)
res["split_in_update_loop"] = {"us_span_unbracketed": round(c["span_us"] or 0, 2), "us_span_bracketed": round(span_s / max(n, 1) * 1e6, 2),
                               "shader_mhz": round(c["shader_mhz"], 1), "workgroup_us": round(c["workgroup_us"], 2),
                               "phase_cycles": {k: round(v) for k, v in c["phase_cycles"].items()},
                               "slab_reduce_us": round(red_us, 2) if red_us else None, "clip_adam_us": round(adam_us, 2) if adam_us else None,
                               "workgroups": _hip.k6_wg_summary(_hip.k6_timing_last_records(False))}
out["k6_standalone"] = res
out["smi_after_k6"] = bench.smi_snapshot()

# ---- HBM: a 1 GiB copy, and the GAE scan at 2048 x 4096
a = th.empty(1 << 28, dtype=th.float32, device=dev)
b = th.empty_like(a)
for _ in range(3):
    b.copy_(a)
th.cuda.synchronize()
e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    b.copy_(a)
e1.record()
th.cuda.synchronize()
out["hbm_copy_GBps"] = round(8.0 * (1 << 28) / (e0.elapsed_time(e1) * 1e-4) / 1e9, 1)
del a, b
print(json.dumps(out))
