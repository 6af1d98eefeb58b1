#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the full PPO actor-learner loop on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one PPO iteration of BASELINE config 4 on every rank: a 32-step vectorised rollout of 4096 synthetic
envs (obs 64, act 8) + value pre-pass + GAE + advantage normalisation + 40 minibatches of 16384 (fused
fwd/bwd, gradient reduce, [RCCL all-reduce when N > 1], clip + Adam).  Inputs are device resident; env shards are
one per GPU (weak scaling).  Rank 0 prints ONE JSON line.

Besides the throughput the line carries
  roofline      dominant kernel of the timed region (ppo_step2_kernel, fp32 MFMA bound), timed with HIP events
                around every launch inside the timed region
  roofline_gae  the GAE scan (HBM bound; the metric's second half): in-loop launches + a size sweep run after
                the timed region (the in-loop 32 x 4096 problem is 2.4 MB, i.e. launch-latency sized)
  cpu_baseline  oracle/torch_port.py (a torch-CPU port of the reference loop) timed on this box's host cores on
                a bounded sample of the same workload (N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak

# BASELINE config 4
N_ENVS, STATE_DIM, ACTION_DIM, HORIZON, BATCH, UPDATE_TIMES, NET_DIMS = 4096, 64, 8, 32, 16384, 40, [128, 128]


def ppo_flops_per_sample(S, h1, h2, A):
    """algorithmic flops of one sample in one minibatch: forward, weight gradients, and input gradients of
    layers 2 and 3 (the input gradient of layer 1 is not needed), actor + critic."""
    def net(out):
        fwd = 2 * (S * h1 + h1 * h2 + h2 * out)
        return fwd + fwd + 2 * (h1 * h2 + h2 * out)
    return net(A) + net(1)


class EventTimer:
    """HIP-event bracket around every call of a wrapped op (events go on torch's current stream, which is the
    stream the kernels are launched on)."""

    def __init__(self):
        self.pairs, self.enabled = [], False

    def wrap(self, fn):
        def inner(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.pairs.append((e0, e1))
            return out
        return inner

    def mean_seconds(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) / max(1, len(self.pairs)) * 1e-3


def gae_sweep(ops, dev):
    out = []
    for H, N in [(32, 4096), (200, 4096), (1024, 4096), (2048, 4096), (4096, 4096), (32, 32768)]:
        g = th.Generator(device=dev).manual_seed(0)
        r, v = th.randn((H, N), device=dev, generator=g), th.randn((H, N), device=dev, generator=g)
        u = th.rand((H, N), device=dev, generator=g) < 0.99
        m = th.rand((H, N), device=dev, generator=g) < 0.995
        nv = th.randn(N, device=dev, generator=g)
        adv, ret = th.empty_like(r), th.empty_like(r)
        run = lambda: ops.gae_scan(r, u, m, v, nv, 0.99, 0.95, mutate=False, adv=adv, ret=ret)   # noqa: E731
        for _ in range(3):
            run()
        th.cuda.synchronize()
        iters = 20
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        th.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / iters
        gbps = 18.0 * H * N / sec / 1e9
        out.append({"H": H, "N": N, "bytes": 18 * H * N, "us": round(sec * 1e6, 2), "GBps": round(gbps, 1),
                    "frac": round(gbps / HBM_PEAK_GBPS, 4)})
    return out


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r02_pmc_traffic.json: separate --pmc
    FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied); None when the file does not carry the kernel."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")))
        for name, v in prof["kernels"].items():
            if name.startswith(kernel_prefix):
                return v["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores() -> int:
    """cores this process may actually run on: affinity mask, capped by the cgroup CPU quota (os.cpu_count() reports
    the whole host and oversubscribing OpenMP threads onto a small quota makes torch-CPU orders of magnitude slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_subprocess(iters: int, timeout_s: int = 240):
    """run the CPU leg in its own process (fresh OpenMP pool, hard timeout) and parse its JSON line."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-iters", str(iters)],
                             capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "env-steps/s", "kind": "port", "error": (out.stderr or "no output")[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "env-steps/s", "kind": "port", "error": f"timed out after {timeout_s}s"}


def cpu_baseline(iters=12):
    """oracle/torch_port.py on the host cores, same workload shape (bounded sample: 1 warm-up + `iters` iterations)."""
    from oracle.torch_port import TorchPortPPO, TorchSynEnv
    th.manual_seed(0)
    cores = usable_cores()
    th.set_num_threads(cores)
    env = TorchSynEnv(N_ENVS, STATE_DIM, ACTION_DIM, 1000, seed=0)
    port = TorchPortPPO(STATE_DIM, ACTION_DIM, tuple(NET_DIMS))
    port.last_state = env.reset()

    def one():
        buf = port.explore(env, HORIZON)
        port.update(list(buf), BATCH, UPDATE_TIMES)

    one()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(N_ENVS * HORIZON * iters / dt, 1), "unit": "env-steps/s", "cores": th.get_num_threads(),
            "kind": "port", "sample": f"{iters} PPO iterations (+1 warm-up) of the same config-4 workload "
                                      f"(4096 envs x 32 steps, 40 minibatches of 16384) via oracle/torch_port.py",
            "seconds": round(dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gae-sweep", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=12)   # ~10-15 s of host work on 16 cores
    ap.add_argument("--k6-sample", type=int, default=16,
                    help="bracket every n-th K6 launch with HIP events (0 = none): each bracket costs ~3 us of stream time")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    opt = ap.parse_args()
    if opt.cpu_baseline_only:
        print(json.dumps(cpu_baseline(opt.cpu_iters)), flush=True)
        return

    from elegantrl_amd import ops, parallel
    from elegantrl_amd.agents import AgentPPO
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config

    rank, world, local_rank = parallel.init_from_env()
    assert world == opt.gpus, f"--gpus {opt.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run for N > 1)"
    local_rank = local_rank % th.cuda.device_count()
    th.cuda.set_device(local_rank)
    dev = th.device(f"cuda:{local_rank}")

    args = Config(AgentPPO, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N_ENVS, "max_step": 1000,
                                        "state_dim": STATE_DIM, "action_dim": ACTION_DIM, "if_discrete": False})
    args.net_dims = list(NET_DIMS)
    args.horizon_len, args.batch_size = HORIZON, BATCH
    args.repeat_times = UPDATE_TIMES * BATCH / HORIZON        # reference formula int(H * repeat_times / B) = 40
    args.gpu_id, args.random_seed = local_rank, 0
    args.world_size, args.rank = world, rank
    th.manual_seed(0)
    agent = AgentPPO(args.net_dims, STATE_DIM, ACTION_DIM, gpu_id=local_rank, args=args)
    parallel.broadcast_(agent._flat)
    env = SynVecEnv(N_ENVS, STATE_DIM, ACTION_DIM, max_step=1000, gpu_id=local_rank, seed=7919 * rank)
    agent.last_state = env.reset()[0]

    from elegantrl_amd import _hip
    t_gae = EventTimer()
    ops.gae_scan = t_gae.wrap(ops.gae_scan)

    def step():
        items = agent.explore_env(env, HORIZON)
        return agent.update_net(list(items))

    log(f"rank {rank}/{world}: agent + env ready, warm-up x{opt.warmup}")
    for _ in range(opt.warmup):
        step()
    log("timed region")
    t_gae.enabled = True
    _hip.k6_timing_enable(opt.k6_sample)         # erl_ppo_step_f32 brackets every n-th K6 launch with HIP events on its stream
    parallel.barrier()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        objs = step()
    th.cuda.synchronize()
    parallel.barrier()
    elapsed = parallel.all_reduce_max_float(time.perf_counter() - t0, device=dev)
    t_gae.enabled = False
    _hip.k6_timing_enable(False)
    k6_seconds, k6_launches = _hip.k6_timing_read()

    log(f"timed region done: {elapsed:.3f}s for {opt.steps} steps")
    if rank != 0:
        return
    env_steps = world * N_ENVS * HORIZON * opt.steps
    flops = ppo_flops_per_sample(STATE_DIM, *NET_DIMS, ACTION_DIM) * BATCH
    ppo_s, n_k6 = (k6_seconds / k6_launches if k6_launches else float("nan")), k6_launches
    gae_s = t_gae.mean_seconds()
    line = {
        "metric": "env_steps_per_sec_ppo_4096envs_obs64", "value": round(env_steps / elapsed, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": round(elapsed / opt.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: AgentPPO, synthetic VecEnv obs_dim=64 act_dim=8, 4096 envs/GPU, "
                               "horizon 32, 40 minibatches x 16384, net [128,128], fp32",
                   "envs_per_gpu": N_ENVS, "horizon": HORIZON, "batch": BATCH, "update_times": UPDATE_TIMES,
                   "parallelism": f"dp{world}" if world > 1 else "single"},
        "roofline": {"kernel": "ppo_step_w4_kernel", "bound": "mfma", "achieved": round(flops / ppo_s / 1e12, 2),
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flops / ppo_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": pmc_traffic("ppo_step_w4_kernel"), "flops_per_launch": flops,
                     "avg_launch_us": round(ppo_s * 1e6, 2), "launches_timed": n_k6},
        "roofline_gae": {"kernel": "gae_exact_kernel (in-loop 32x4096)", "bound": "hbm",
                         "achieved": round(18.0 * HORIZON * N_ENVS / gae_s / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(18.0 * HORIZON * N_ENVS / gae_s / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                         "bytes_per_launch": 18 * HORIZON * N_ENVS, "avg_launch_us": round(gae_s * 1e6, 2)},
        "objectives_last": [round(float(x), 6) for x in objs],
    }
    if not opt.no_gae_sweep:
        log("GAE size sweep")
        sweep = gae_sweep(ops, dev)
        line["roofline_gae"]["sweep"] = sweep
        big = next(x for x in sweep if (x["H"], x["N"]) == (2048, 4096))
        line["roofline_gae"]["at_2048x4096"] = {"kernel": "gae_lookback_kernel (+ slot memset)", "achieved": big["GBps"],
                                                "frac": big["frac"], "us": big["us"], "bytes_per_launch": big["bytes"],
                                                "traffic": pmc_traffic("gae_lookback_kernel")}
    if world == 1 and not opt.no_cpu_baseline:
        log(f"cpu baseline ({usable_cores()} usable cores of {os.cpu_count()})")
        line["cpu_baseline"] = cpu_baseline_subprocess(opt.cpu_iters)
    log("done")
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
    from elegantrl_amd import parallel as _parallel
    _parallel.shutdown()      # every rank: barrier -> RCCL communicator -> process group (rank 0 has printed its line by now)
