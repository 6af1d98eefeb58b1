#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the full PPO actor-learner loop on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one PPO iteration of BASELINE config 4 on every rank: ONE launch for the 32-step vectorised rollout of 4096
synthetic envs (obs 64, act 8) + value pre-pass + GAE + the advantage-normalisation sums, then 40 minibatches of 16384
(fused fwd/bwd, gradient-slab reduce [+ the data-parallel exchange inside that launch, or an RCCL all-reduce, when N > 1],
clip + Adam).  Inputs are device resident; env shards are one per GPU (weak scaling).  Rank 0 prints ONE JSON line
(exactly one line on stdout: libraries' C-level stdout is redirected to stderr).

Besides the throughput the line carries
  roofline      dominant kernel of the timed region (the PPO minibatch kernel: ppo_step_s3_kernel, bf16 matrix pipe with
                fp32-equivalent split arithmetic, or ppo_step_w4_kernel on the fp32 MFMA).  `avg_launch_us` is the kernel's
                own span on the device clock (first workgroup in -> last workgroup out), measured on every launch of the
                timed region; `event_bracket_us` the HIP-event bracket around the same launches, `event_bracket_null_us`
                that bracket around an empty launch, `kernel_us_rocprof` the committed rocprofv3 average it must agree with
  roofline_gae  the GAE scan (HBM bound; the metric's second half): a size sweep run after the timed region (the loop
                itself has no GAE launch left when the rollout's epilogue computes it; `--config` runs that do report them)
  breakdown     per-stage GPU time of one iteration (rollout / minibatch kernel / tail / rest), with `consistent`
  cpu_baseline  oracle/cpu_baseline.py (a torch-CPU port of the reference loop, validated against the reference's own
                classes: profiles/r04_cpu_baseline_reference_vs_port_*.json) timed per stage on this box's host cores on a
                bounded sample of the same workload (N = 1 only)
  extra.per_rank_ms_per_step   (N > 1) fastest / slowest rank's own time per step
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
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak (/opt/skills/guides/MI355X_MICROARCH.md)
SPLIT_TERMS = 6  # bf16 partial products per fp32-equivalent product in the split-arithmetic K6 (csrc/ppo_step_s3_impl.h)

# BASELINE configs (SURVEY.md section 8 header / 8d).  c4 = configs[3] is the metric's configuration and the default; the
# others are run with --config and recorded under profiles/ (same JSON schema, the workload named in config.workload).
PPO_CONFIGS = {
    "c4": dict(metric="env_steps_per_sec_ppo_4096envs_obs64", env="syn", N=4096, S=64, A=8, H=32, B=16384, update_times=40,
               net_dims=[128, 128], hyper={},
               workload="BASELINE configs[3]: AgentPPO, synthetic VecEnv obs_dim=64 act_dim=8, 4096 envs/GPU, horizon 32, "
                        "40 minibatches x 16384, net [128,128], fp32"),
    "c5": dict(metric="env_steps_per_sec_ppo_ant_shaped_8192envs", env="syn", N=8192, S=60, A=8, H=32, B=16384, update_times=80,
               net_dims=[128, 128], hyper=dict(learning_rate=5e-4, lambda_entropy=0.0, reward_scale=0.01),
               workload="BASELINE configs[4]: AgentPPO, Isaac-Gym-Ant-shaped synthetic VecEnv obs_dim=60 act_dim=8, 8192 envs/GPU, "
                        "horizon 32, batch 16384, 5 passes = 80 minibatches, lr 5e-4, entropy 0 (examples/plan_Isaac_Gym.py:36-64), "
                        "net [128,128], fp32"),
    "c2": dict(metric="env_steps_per_sec_ppo_pendulum_4096envs", env="pendulum", N=4096, S=3, A=1, H=200, B=16384, update_times=40,
               net_dims=[128, 64], hyper=dict(gamma=0.97, reward_scale=0.25, learning_rate=4e-4),
               workload="BASELINE configs[1]: AgentPPO, GPU-resident vectorised Pendulum-v1, 4096 envs, horizon 200, "
                        "40 minibatches x 16384, net [128,64], fp32"),
    # SURVEY.md 8(d) "also report the reference-default shape": the reference's own on-policy defaults (elegantrl/train/config.py:55-58:
    # batch_size 128, horizon_len 2048, repeat_times 8.0 => int(2048 * 8 / 128) = 128 minibatches of 128) at config 4's env shape.  The
    # only BASELINE-adjacent shape at which the GAE scan runs IN THE LOOP at 2048 x 4096 (fused_gae off: the look-back kernel as a launch
    # of its own, `roofline_gae` = its in-loop figure); the rollout (2048 steps in one launch) is the dominant kernel here.
    "cd": dict(metric="env_steps_per_sec_ppo_4096envs_obs64_reference_default_shape", env="syn", N=4096, S=64, A=8, H=2048, B=128,
               update_times=128, net_dims=[128, 128], hyper=dict(fused_gae=False),
               workload="reference-default on-policy shape (elegantrl/train/config.py:55-58: horizon_len 2048, batch_size 128, repeat_times 8 => "
                        "128 minibatches x 128) on BASELINE configs[3]'s synthetic VecEnv obs_dim=64 act_dim=8, 4096 envs/GPU, net [128,128], fp32"),
    # BASELINE configs[0] (helloworld/helloworld_PPO_single_file.py on Pendulum-v1, num_envs = 4: plumbing): the helloworld's hyper-parameters
    # (:535-553: net_dims [64, 32], gamma 0.97, repeat_times 16; Config defaults horizon_len 2048, batch_size 128 => 256 minibatches) on the
    # GPU-resident Pendulum with FOUR envs.  Launch-bound by construction (one 16-env rollout workgroup, two minibatch workgroups): the line
    # documents what the plumbing case costs on a GPU; `tests/test_compat.py::test_config0_helloworld_shaped_pendulum_run_with_four_envs`
    "c1": dict(metric="env_steps_per_sec_ppo_pendulum_4envs_helloworld_shape", env="pendulum", N=4, S=3, A=1, H=2048, B=128, update_times=256,
               net_dims=[64, 32], hyper=dict(gamma=0.97),
               workload="BASELINE configs[0]: helloworld_PPO_single_file.py's Pendulum hyper-parameters (net [64,32], gamma 0.97, repeat_times 16, "
                        "horizon 2048, batch 128 => 256 minibatches) on the GPU-resident Pendulum-v1 with num_envs = 4 (plumbing; the reference runs it on CPU torch)"),
    # not a measurement: the REHEARSAL shape of the N > 1 code path (tests/test_bench_gpu.py runs `--gpus 8` with eight ranks SHARING the one
    # GPU of the test box over gloo: barrier / max-over-ranks timing, route probe + self-test + selection, the exchange inside the loop,
    # the all-reduce micro-benchmark, per-rank times) -- small enough that eight ranks' kernels and their exchange workgroups fit one device
    "cr": dict(metric="rehearsal_env_steps_per_sec_ppo_512envs", env="syn", N=512, S=8, A=2, H=16, B=2048, update_times=4, net_dims=[64, 64],
               hyper={}, workload="rehearsal of the data-parallel path (not a BASELINE configuration): 512 envs/rank, obs 8, act 2, horizon 16, "
                                  "4 minibatches x 2048, net [64,64], fp32"),
    # not a BASELINE configuration: the network of the reference's LunarLanderContinuous demo (examples/demo_A2C_PPO.py:117,
    # net_dims (256, 128), with its hyper-parameters :118-125) on a synthetic VecEnv of that env's shape, vectorised like configs[3];
    # its minibatch loop runs on csrc/ppo_step_wd_impl.h (rollout and value pre-pass on the layered erl_mlpn_* path)
    "cw": dict(metric="env_steps_per_sec_ppo_net256x128_lunarlander_shaped_4096envs", env="syn", N=4096, S=8, A=2, H=32, B=16384,
               update_times=40, net_dims=[256, 128],
               hyper=dict(gamma=0.99, reward_scale=0.5, learning_rate=2e-4, lambda_gae_adv=0.97, lambda_entropy=0.04),
               workload="reference demo network net_dims (256,128) (examples/demo_A2C_PPO.py:117-125 hyper-parameters) on a "
                        "LunarLanderContinuous-shaped synthetic VecEnv obs_dim=8 act_dim=2, 4096 envs/GPU, horizon 32, 40 minibatches x 16384, fp32"),
}
N_ENVS, STATE_DIM, ACTION_DIM, HORIZON, BATCH, UPDATE_TIMES, NET_DIMS = 4096, 64, 8, 32, 16384, 40, [128, 128]


def select_config(name: str):
    """point the module-level workload constants at one of the PPO configs"""
    global N_ENVS, STATE_DIM, ACTION_DIM, HORIZON, BATCH, UPDATE_TIMES, NET_DIMS
    c = PPO_CONFIGS[name]
    N_ENVS, STATE_DIM, ACTION_DIM, HORIZON, BATCH, UPDATE_TIMES, NET_DIMS = (c["N"], c["S"], c["A"], c["H"], c["B"],
                                                                              c["update_times"], list(c["net_dims"]))
    return c


def ppo_flops_per_sample(S, h1, h2, A):
    """algorithmic flops of one sample in one minibatch: forward, weight gradients, and input gradients of
    layers 2 and 3 (the input gradient of layer 1 is not needed), actor + critic."""
    def net(out):
        fwd = 2 * (S * h1 + h1 * h2 + h2 * out)
        return fwd + fwd + 2 * (h1 * h2 + h2 * out)
    return net(A) + net(1)


def rollout_flops_per_env_step(S, h1, h2, A):
    """algorithmic flops of one env-step inside the persistent rollout kernel: actor forward + critic forward (the value pre-pass
    rides in the same launch) + the synthetic env's s Ws + a Wa"""
    return 2 * (S * h1 + h1 * h2 + h2 * A) + 2 * (S * h1 + h1 * h2 + h2) + 2 * (S * S + A * S)


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

    def each_ms(self):
        return [round(a.elapsed_time(b), 3) for a, b in self.pairs]


def gae_sweep(ops, dev):
    """the GAE scan over the SURVEY 8d sizes, two clocks per size: `kernel_us` = the kernel's own first-workgroup-in to last-workgroup-out
    span on the device clock (erl_kernel_span_*: what rocprofv3's kernel duration measures) and `call_us` = HIP events around 20
    back-to-back ops.gae_scan calls (interpreter + launch + kernel: at the small sizes mostly not the kernel).  GB/s and frac are the
    KERNEL's (18 algorithmic bytes per element / kernel_us)."""
    from elegantrl_amd import _hip
    out = []
    flush = None
    for H, N in [(32, 4096), (128, 4096), (200, 4096), (1024, 4096), (2048, 4096), (4096, 4096), (32, 32768)]:   # SURVEY 8d sizes (+ 4096 x 4096)
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
        _hip.kernel_span_enable(True)
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        th.cuda.synchronize()
        kernel_us, n_k = _hip.kernel_span_read(_hip.SPAN_GAE)
        _hip.kernel_span_enable(False)
        # COLD: the back-to-back calls above re-read the same inputs, and 84 MB of them (2048 x 4096) fit the 256 MB Infinity Cache -- that
        # figure is the kernel on warm inputs.  Here a 640 MB buffer is rewritten between calls, so every call finds its inputs in HBM only,
        # as the scan does in a training loop behind a rollout's writes (`--config cd` measures exactly that, in the loop).
        if flush is None:
            flush = th.zeros(640 << 20, dtype=th.uint8, device=dev)
        _hip.kernel_span_enable(True)
        for _ in range(8):
            flush.add_(1)
            run()
        th.cuda.synchronize()
        cold_us, n_c = _hip.kernel_span_read(_hip.SPAN_GAE)
        _hip.kernel_span_enable(False)
        call_s = e0.elapsed_time(e1) * 1e-3 / iters
        sec = kernel_us * 1e-6 if kernel_us else call_s
        gbps = 18.0 * H * N / sec / 1e9
        out.append({"H": H, "N": N, "bytes": 18 * H * N, "kernel_us": round(kernel_us, 2) if kernel_us else None,
                    "cold_kernel_us": round(cold_us, 2) if cold_us else None, "call_us": round(call_s * 1e6, 2),
                    "us": round(sec * 1e6, 2), "GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4),
                    "cold_GBps": round(18.0 * H * N / cold_us / 1e3, 1) if cold_us else None,
                    "cold_frac": round(18.0 * H * N / cold_us / 1e3 / HBM_PEAK_GBPS, 4) if cold_us else None,
                    "inputs": "kernel_us / frac: warm (the same inputs again and again: Infinity Cache); cold_*: a 640 MB buffer rewritten between calls",
                    "call_GBps": round(18.0 * H * N / call_s / 1e9, 1)})
    return out


# the files a kernel's HBM traffic depends on (what the PMC passes were collected on is stamped with their hash)
KERNEL_SOURCES = {
    "ppo_step_w4_kernel": ["ppo_step_w4_impl.h", "ppo_step_w4.hip", "ppo_step.h", "mlp_chain.h", "mlp_tiles.h", "ppo_objective.h"],
    "ppo_step_s3_kernel": ["ppo_step_s3_impl.h", "split_bf16.h", "ppo_step_s3.hip", "ppo_step_s3_pre.hip", "s3_image.h", "ppo_step_w4_impl.h", "ppo_step.h",
                           "mlp_chain.h", "mlp_tiles.h", "ppo_objective.h"],
    "ppo_step2_kernel": ["ppo_step.hip", "ppo_step.h", "mlp_chain.h", "mlp_tiles.h", "ppo_objective.h"],
    "gae_lookback_kernel": ["gae_lookback.hip"],
    "ppo_step_wd_kernel": ["ppo_step_wd_impl.h", "ppo_step_wd.hip", "ppo_step_wd.h", "ppo_step_s3_impl.h", "split_bf16.h", "s3_image.h",
                           "ppo_step_w4_impl.h", "ppo_step.h", "mlp_chain.h", "mlp_tiles.h", "ppo_objective.h"],
}
PMC_FILE = os.path.join("profiles", "r06_pmc_traffic.json")
KTIME_FILE = os.path.join("profiles", "r06_kernel_times.json")     # tools/kstats_summarise.py over rocprofv3 --kernel-trace --stats of this command


def kernel_source_sha16(kernel: str):
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES.get(kernel, []):
        with open(os.path.join(ROOT, "elegantrl_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


C3_PMC_FILE = os.path.join("profiles", "r06_c3_pmc.json")            # tools/c3_pmc_workload.py under rocprofv3 --pmc (critic_tile_kernel, replay_sample_kernel)
K9_PMC_FILE = os.path.join("profiles", "r06_k9_pmc_by_size.json")    # ... replay_sample_kernel per (num_seqs, B) case: FETCH_SIZE / WRITE_SIZE bytes
WIDE_PMC_FILE = os.path.join("profiles", "r04_wide_pmc_traffic_S8_h128.json")     # tools/wide_pmc_workload.py, WD_S=8 WD_A=2 WD_ONLY=128 (cw's shape)


def pmc_traffic(kernel, file=None):
    """(HBM bytes per launch, provenance) from the committed rocprofv3 PMC passes (PMC_FILE: separate --pmc FETCH_SIZE /
    WRITE_SIZE passes, gfx950 correction applied, tools/pmc_summarise.py).  The file stamps every kernel with the hash of
    the sources it was collected on; when the kernel's sources have changed since, the number is STALE and None is
    reported (bench.py itself cannot collect counters: that needs rocprofv3 around the process)."""
    file = file or PMC_FILE
    src = {"file": file, "kernel_source_sha16": kernel_source_sha16(kernel), "collected_on_sha16": None, "stale": True}
    try:
        prof = json.load(open(os.path.join(ROOT, file)))
        v = prof["kernels"][kernel]
        src["collected_on_sha16"] = v.get("source_sha16")
        src["stale"] = v.get("source_sha16") != src["kernel_source_sha16"]
        return (None if src["stale"] else v["hbm_bytes_per_launch"]), src
    except Exception:
        return None, src


def rocprof_kernel_us(kernel):
    """(average kernel duration under rocprofv3 --kernel-trace of this bench command on the authoring round's box, provenance) from the
    committed summary KTIME_FILE; None when the kernel's sources have changed since it was collected.  A cross-check for
    `avg_launch_us`, which is measured live: the two must agree."""
    src = {"file": KTIME_FILE, "kernel_source_sha16": kernel_source_sha16(kernel), "collected_on_sha16": None, "stale": True}
    try:
        v = json.load(open(os.path.join(ROOT, KTIME_FILE)))["kernels"][kernel]
        src["collected_on_sha16"] = v.get("source_sha16")
        src["stale"] = v.get("source_sha16") != src["kernel_source_sha16"]
        src["calls"] = v.get("calls")
        return (None if src["stale"] else v["avg_us"]), src
    except Exception:
        return None, src


SHADER_PEAK_MHZ = 2400.0      # the clock the dense-MFMA peaks are quoted at (MI355X_MICROARCH.md: max clock)


def smi_snapshot(timeout_s: int = 20):
    """what the box says about itself right after the timed region (clocks, power cap / draw, temperature, partition modes): the pool's
    boxes do not all run this workload at the same speed (BENCH_r02..r04: the driver's box ran the minibatch kernel 15-40 % slower than
    the boxes the kernel was tuned on), and the bench line should carry the evidence.  Best effort: whichever of rocm-smi / amd-smi
    answers within the timeout; only keys about clocks / power / temperature / partitioning are kept."""
    import re
    import subprocess
    keep = re.compile(r"sclk|mclk|fclk|gfxclk|uclk|gfx_0\.clk\.value|mem_0\.clk\.value|power|temperature_hotspot|temperature_mem|junction|perf|partition|throttl|"
                      r"vbios|model|sys\.current_frequency|frequency_levels", re.I)
    drop = re.compile(r"N/A|vclk|dclk|socclk|deep_sleep|clk_locked|min_clk|max_clk|shutdown|slowdown|ppt1|\.unit$", re.I)
    out = {}
    for cmd in (["rocm-smi", "-a", "--json"], ["amd-smi", "static", "--json"], ["amd-smi", "metric", "--json"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            txt = r.stdout.strip()
            if not txt:
                continue
            starts = [i for i in (txt.find("{"), txt.find("[")) if i >= 0]
            data = json.loads(txt[min(starts):])
        except Exception as e:      # tool missing / no JSON / timeout: say so, go on
            out[" ".join(cmd)] = {"error": repr(e)[:120]}
            continue
        flat = {}

        def walk(prefix, v):
            if isinstance(v, dict):
                for k, x in v.items():
                    walk(f"{prefix}.{k}" if prefix else str(k), x)
            elif isinstance(v, list):
                for i, x in enumerate(v[:2]):          # (one GPU is visible; keep the first entries only)
                    walk(f"{prefix}[{i}]", x)
            elif keep.search(prefix) and not drop.search(prefix) and not drop.search(str(v)) and len(flat) < 40:
                flat[prefix] = v
        walk("", data)
        out[" ".join(cmd)] = flat
    return out


_JSON_FD = None


def claim_stdout():
    """ONE JSON line on stdout is the contract: libraries that print to the C-level stdout (RCCL's version banner at communicator
    creation, flushed at exit, i.e. AFTER the line) must not end up there.  File descriptor 1 is pointed at stderr for the whole run;
    the result line is written to the saved descriptor by emit()."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line: dict):
    data = (json.dumps(line) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def self_launch(n: int):
    """`python bench.py --gpus N` with no launcher around it (no WORLD_SIZE in the environment): start the N ranks ourselves, exactly as the
    driver's documented form does -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <the same arguments>` -- on a free port, and hand its stdout (rank 0's ONE JSON line) and exit code through.  The reference's
    analog is train_agent_multiprocessing_multi_gpu starting its Learner / Worker processes itself (elegantrl/train/run.py:165-202)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"--gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}")
    sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))


def quiet_gc():
    """after the warm-up: collect once and move what survives (everything torch built at import) to the permanent generation, as
    elegantrl_amd.train.run does after its first iterations: a full collection of the interpreter costs 40-65 ms of idle GPU in
    the middle of a 50 ms timed region (measured: tools/iter_times.py), and which step it lands on depends on allocation counts"""
    import gc
    gc.collect()
    gc.freeze()


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


def cpu_baseline_subprocess(iters: int, config: str = "c4", timeout_s: int = 300):
    """run the CPU leg in its own process (fresh OpenMP pool, hard timeout, and -- where /root/reference is mounted -- the
    reference's own `elegantrl` package instead of this repository's import alias) and parse its JSON line."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-iters", str(iters),
                              "--config", config],
                             capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "env-steps/s", "kind": "port", "error": (out.stderr or "no output")[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "env-steps/s", "kind": "port", "error": f"timed out after {timeout_s}s"}


def cpu_baseline(iters=12, config="c4"):
    """the reference's CPU path on this box's host cores, same workload, bounded sample (oracle/cpu_baseline.py): the
    reference's own classes where /root/reference is mounted (kind "reference", with the torch port timed right after it on the
    same cores as `port_same_cores`, so that the port's number on a box without the reference is a validated stand-in), else the
    port (kind "port").  Per-stage seconds: explore_env / get_advantages / update_net (+ buffer_update / sample for c3)."""
    from oracle import cpu_baseline as cb
    cores = usable_cores()
    th.set_num_threads(cores)
    if config == "c3":
        kw = dict(N=64, S=11, A=3, H=64, B=256, updates=64, net_dims=[256, 256], max_size=1_000_000 // 64, iters=iters)
        run = cb.sac
    else:
        c = PPO_CONFIGS[config]
        kw = dict(N=c["N"], S=c["S"], A=c["A"], H=c["H"], B=c["B"], update_times=c["update_times"], net_dims=c["net_dims"], iters=iters,
                  hyper=c["hyper"], max_step=200 if c["env"] == "pendulum" else 1000, env_kind=c["env"])
        run = cb.ppo
    if cb.reference_available():
        line = run("reference", **kw)
        port = run("port", **kw)
        line["port_same_cores"] = {k: port[k] for k in ("value", "unit", "seconds", "stage_seconds")}
        line["port_over_reference"] = round(port["value"] / line["value"], 3)
    else:
        line = run("port", **kw)
    return line


def bench_sac(opt):
    """--config c3: BASELINE configs[2], AgentSAC on a Hopper-v3-shaped synthetic env (obs 11, act 3) with a FULL 1e6-transition
    replay ring.  One step = one off-policy iteration as elegantrl/train/run.py drives it: explore_env (64 steps x 64 envs) ->
    buffer.update -> update_net = 64 x [ReplayBuffer.sample(256) + the SAC update].  value = SAC updates/s; `roofline` is the
    sample kernel K9 (HBM bound, (2S + A + 3) * 4 B read + the same written + 8 B id per sample)."""
    from elegantrl_amd import ops
    from elegantrl_amd.agents import AgentSAC
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config, ReplayBuffer
    assert opt.gpus == 1, "config 3 is a single-GPU configuration"
    dev = th.device("cuda:0")
    N, S, A, H, B, UPD, NET = 64, 11, 3, 64, 256, 64, [256, 256]
    max_size = 1_000_000 // N                                           # ring capacity in time rows: 1e6 transitions
    args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A,
                                        "if_discrete": False})
    args.net_dims, args.horizon_len, args.batch_size = NET, H, B
    args.repeat_times = UPD * B / max_size                              # AgentBase.update_net: int(cur_size * repeat_times / B) = 64
    args.gpu_id, args.random_seed = 0, 0
    th.manual_seed(0)
    agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
    env = SynVecEnv(N, S, A, max_step=1000, gpu_id=0, seed=0)
    agent.last_state = env.reset()[0]
    buf = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N)
    g = th.Generator(device=dev).manual_seed(1)
    for _ in range(2):                                                   # fill the ring completely (and wrap once)
        buf.update((th.randn((max_size // 2 + 7, N, S), device=dev, generator=g), th.randn((max_size // 2 + 7, N, A), device=dev, generator=g).tanh(),
                    th.randn((max_size // 2 + 7, N), device=dev, generator=g), th.rand((max_size // 2 + 7, N), device=dev, generator=g) < 0.99,
                    th.rand((max_size // 2 + 7, N), device=dev, generator=g) < 0.995))
    assert buf.if_full and buf.cur_size == max_size
    t_k9 = EventTimer()
    buf.sample = t_k9.wrap(buf.sample)        # (the class's sample: the interleaved ring's row kernel since round 6)

    def step():
        buf.update(agent.explore_env(env, H))
        return agent.update_net(buf)

    from elegantrl_amd import _hip
    for _ in range(opt.warmup):
        step()
    quiet_gc()
    t_k9.enabled = True
    _hip.kernel_span_enable(True)      # the critic's training pass and the sample kernel leave their own device-clock spans (no brackets)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        objs = step()
    th.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t_k9.enabled = False
    crit_us, crit_n = _hip.kernel_span_read(_hip.SPAN_SAC_CRITIC_TRAIN)
    k9_loop_us, k9_loop_n = _hip.kernel_span_read(_hip.SPAN_REPLAY_SAMPLE)
    _hip.kernel_span_enable(False)
    sample_in_step = len(t_k9.pairs) == 0      # (update_net handed ring + ids to the step: the gather rides in the step's first launch)
    if sample_in_step:                         # the sample kernel at the loop's batch size, on its own, for the roofline object
        t_k9.enabled = True
        for _ in range(200):
            buf.sample(B, reuse=True)
        th.cuda.synchronize()
        t_k9.enabled = False
    k9_s = t_k9.mean_seconds()
    bytes_per = (2 * (2 * S + A + 3) * 4 + 8) * B
    # the sample kernel's own span at B in {256, 4096, 2^20} on this ring (num_seqs = 64) and on a one-env ring of the same capacity
    # (num_seqs = 1: BASELINE.md section 3), 18 algorithmic... (2S + A + 3) * 4 B read + the same written + 8 B id per sample
    k9_sizes = []
    one = ReplayBuffer(max_size=1_000_000, state_dim=S, action_dim=A, gpu_id=0, num_seqs=1)
    one.update((th.randn((999_999, 1, S), device=dev, generator=g), th.randn((999_999, 1, A), device=dev, generator=g).tanh(),
                th.randn((999_999, 1), device=dev, generator=g), th.rand((999_999, 1), device=dev, generator=g) < 0.99,
                th.rand((999_999, 1), device=dev, generator=g) < 0.995))
    for ring, seqs in ((buf, N), (one, 1)):
        for bsz in (256, 4096, 1 << 20):
            idx = th.randint((ring.cur_size - 1) * seqs, (bsz,), device=dev, generator=g)
            for _ in range(3):
                ring.sample(bsz, ids=idx, reuse=True)
            th.cuda.synchronize()
            _hip.kernel_span_enable(True)
            for _ in range(20):
                ring.sample(bsz, ids=idx, reuse=True)
            us, _n = _hip.kernel_span_read(_hip.SPAN_REPLAY_SAMPLE)
            _hip.kernel_span_enable(False)
            by = (2 * (2 * S + A + 3) * 4 + 8) * bsz
            k9_sizes.append({"num_seqs": seqs, "B": bsz, "bytes": by, "kernel_us": round(us, 2) if us else None,
                             "GBps": round(by / (us * 1e-6) / 1e9, 1) if us else None,
                             "frac": round(by / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if us else None})
    del one
    # the dominant in-loop kernel of the step: the critic ensemble's training pass (forward of the shared encoder + E decoders, loss
    # gradient, backward to the encoder output; weight gradients are dw_table's): algorithmic flops per launch
    E_, h0, h1 = agent.num_ensembles, NET[0], NET[1]
    crit_flops = 2 * B * ((S + A) * h0 + E_ * (h0 * h1 + h1) + E_ * (h1 + h1 * h0))
    crit_tr, crit_src = None, None
    try:
        prof = json.load(open(os.path.join(ROOT, C3_PMC_FILE)))["kernels"]["critic_tile_kernel"]
        crit_tr = prof.get("hbm_bytes_per_launch")
        crit_src = {"file": C3_PMC_FILE, "mfma_busy_frac": prof.get("mfma_busy_frac_of_kernel_time_at_2.4GHz"), "avg_duration_us": prof.get("avg_duration_us"),
                    "note": "all three passes of critic_tile_kernel averaged (target on next_state / training / target on the policy-gradient sample)"}
    except Exception:
        pass
    # the kernel's capability away from the launch floor: one sample call of 2^20 transitions on the same ring
    big = th.randint((max_size - 1) * N, (1 << 20,), device=dev, generator=g)
    t_k9.enabled = False
    for _ in range(3):
        buf.sample(1 << 20, ids=big, reuse=True)
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        buf.sample(1 << 20, ids=big, reuse=True)
    e1.record()
    th.cuda.synchronize()
    big_s = e0.elapsed_time(e1) * 1e-4
    big_bytes = (2 * (2 * S + A + 3) * 4 + 8) * (1 << 20)
    line = {
        "metric": "sac_updates_per_sec_replay1M_batch256", "value": round(UPD * opt.steps / elapsed, 1), "unit": "updates/s",
        "n_gpus": 1, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": round(elapsed / opt.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: AgentSAC, Hopper-v3-shaped synthetic VecEnv obs_dim=11 act_dim=3, 64 envs, full "
                               "ReplayBuffer of 1e6 transitions, per step 64x64 env steps + 64 x [sample(256) + SAC update], "
                               "net [256,256], 4 critics, fp32", "name": "c3", "envs_per_gpu": N, "horizon": H, "batch": B,
                   "update_times": UPD, "parallelism": "single"},
        "env_steps_per_sec": round(N * H * opt.steps / elapsed, 1),
        "us_per_update": round(elapsed / opt.steps / UPD * 1e6, 1),
        "roofline": ({"kernel": "critic_tile_kernel<1> (the critic ensemble's training pass: dominant in-loop kernel of the fused SAC step)",
                      "bound": "mfma", "achieved": round(crit_flops / (crit_us * 1e-6) / 1e12, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": round(crit_flops / (crit_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "traffic": crit_tr, "traffic_source": crit_src,
                      "flops_per_launch": crit_flops, "avg_launch_us": round(crit_us, 2), "launches_timed": crit_n, "in_loop": True,
                      "timer": "the kernel's own span on the device clock (first workgroup in to last workgroup out), every launch of the timed region",
                      "note": "fp32 MFMA 16x16x4; a 16-sample tile x decoder per workgroup streams the decoder's whole 256 x 256 layer (forward and "
                              "transposed) through ONE CU: bound by that CU's ~12 B/clk from L2, not by the matrix pipe or HBM (DESIGN.md section 4, SAC)"}
                     if crit_us else None),
        "roofline_sample": {"kernel": "replay_sample_rows_kernel (the interleaved ring, round 6: one row [state | action | reward | undone | unmask] per transition, "
                                      "its next state the head of the following row)", "bound": "hbm", "achieved": round(bytes_per / k9_s / 1e9, 2), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(bytes_per / k9_s / 1e9 / HBM_PEAK_GBPS, 5), "traffic": None,
                     "bytes_per_launch": bytes_per, "avg_launch_us": round(k9_s * 1e6, 2), "launches_timed": len(t_k9.pairs),
                     "in_loop": not sample_in_step,
                     "note": ("the loop's sample rides inside the SAC step's first launch (erl_sac_update_ring_f32): the kernel was timed on its own, "
                              "200 calls at the loop's batch size" if sample_in_step else "timed in the loop"),
                     "at_batch_2^20": {"bytes_per_launch": big_bytes, "us": round(big_s * 1e6, 1),
                                       "achieved": round(big_bytes / big_s / 1e9, 1), "frac": round(big_bytes / big_s / 1e9 / HBM_PEAK_GBPS, 4)},
                     "kernel_span_by_size": k9_sizes, "traffic_by_size_file": K9_PMC_FILE},
        "objectives_last": [round(float(x), 6) for x in objs],
    }
    if not opt.no_cpu_baseline:
        log(f"cpu baseline ({usable_cores()} usable cores of {os.cpu_count()})")
        line["cpu_baseline"] = cpu_baseline_subprocess(max(2, opt.cpu_iters // 2), "c3")
    emit(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gae-sweep", action="store_true")
    ap.add_argument("--gc-after-warmup", action="store_true", help="gc.collect() + gc.freeze() and the null-bracket measurement between the warm-up and "
                                                                     "the timed region (rounds 3-5) instead of before the warm-up with the collector off for the region")
    ap.add_argument("--gae-sweep-after", action="store_true", help="run the GAE size sweep after the timed region (rounds 1-5) instead of before the warm-up")
    ap.add_argument("--no-smi", action="store_true", help="skip the rocm-smi / amd-smi snapshot after the timed region (`clocks.smi`)")
    ap.add_argument("--cpu-iters", type=int, default=12)   # ~10-15 s of host work on 16 cores
    ap.add_argument("--k6-sample", type=int, default=17,
                    help="of every n K6 launches one sits in a HIP-event bracket and one more leaves its per-workgroup records without a bracket "
                         "(0 = none); a period coprime with the update loop's length samples every position of the loop equally often -- with 16 "
                         "and 40 launches per loop every fifth sample was the loop's FIRST launch, which finds the instruction caches cold")
    ap.add_argument("--repeats", type=int, default=5,
                    help="after the primary timed region, repeat it this many times and report min / median / max ms per step "
                         "(`extra`; box variance next to the one primary sample; 0 = off)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--eager-logs", action="store_true", help="read update_net's logged objectives at once (one host sync per iteration with the GPU "
                                                              "idle behind it) instead of one rollout late, as train_agent does by default")
    ap.add_argument("--config", choices=["c4", "c1", "c2", "c3", "c5", "cw", "cd", "cr"], default="c4",
                    help="BASELINE configuration: c4 = configs[3] (the metric; default), c2 = Pendulum 4096 envs, "
                         "c3 = SAC on a 1e6-transition ring, c5 = Ant-shaped 8192 envs, cw = the reference demo's (256,128) network, "
                         "cd = the reference's default horizon / batch shape (2048 x 4096 rollout, 128 minibatches of 128)")
    opt = ap.parse_args()
    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(opt.gpus)
    claim_stdout()
    if opt.cpu_baseline_only:
        emit(cpu_baseline(opt.cpu_iters, opt.config))
        return
    if opt.config == "c3":
        return bench_sac(opt)
    cfg = select_config(opt.config)

    from elegantrl_amd import ops, parallel
    from elegantrl_amd.agents import AgentPPO
    from elegantrl_amd.envs import PendulumVecEnv, SynVecEnv
    from elegantrl_amd.train import Config

    rank, world, local_rank = parallel.init_from_env()
    assert world == opt.gpus, f"--gpus {opt.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run for N > 1)"
    local_rank = local_rank % th.cuda.device_count()
    th.cuda.set_device(local_rank)
    dev = th.device(f"cuda:{local_rank}")

    pendulum = cfg["env"] == "pendulum"
    max_step = 200 if pendulum else 1000
    args = Config(AgentPPO, PendulumVecEnv if pendulum else SynVecEnv,
                  {"env_name": "Pendulum-v1" if pendulum else "SynVecEnv", "num_envs": N_ENVS, "max_step": max_step,
                   "state_dim": STATE_DIM, "action_dim": ACTION_DIM, "if_discrete": False})
    for k, v in cfg["hyper"].items():
        setattr(args, k, v)
    args.net_dims = list(NET_DIMS)
    args.horizon_len, args.batch_size = HORIZON, BATCH
    args.repeat_times = UPDATE_TIMES * BATCH / HORIZON        # reference formula int(H * repeat_times / B) = 40
    args.gpu_id, args.random_seed = local_rank, 0
    args.world_size, args.rank = world, rank
    th.manual_seed(0)
    agent = AgentPPO(args.net_dims, STATE_DIM, ACTION_DIM, gpu_id=local_rank, args=args)
    parallel.broadcast_(agent._flat)
    env = (PendulumVecEnv(N_ENVS, max_step=max_step, gpu_id=local_rank, seed=7919 * rank) if pendulum else
           SynVecEnv(N_ENVS, STATE_DIM, ACTION_DIM, max_step=max_step, gpu_id=local_rank, seed=7919 * rank))
    agent.last_state = env.reset()[0]

    from elegantrl_amd import _hip
    t_gae = EventTimer()
    ops.gae_scan = t_gae.wrap(ops.gae_scan)

    # one event bracket per call around the two drop-in entry points (2 per iteration: ~6 us of a 2.4 ms step): the parts of a step
    t_explore, t_update = EventTimer(), EventTimer()
    explore_env, update_net = t_explore.wrap(agent.explore_env), t_update.wrap(agent.update_net)

    # the loop of elegantrl_amd.train.run.train_agent_single_process: an update's three logged objectives are read ONE ROLLOUT LATE
    # (update_net(lazy=True) -> PendingLogs), i.e. after the next rollout has been enqueued -- the GPU runs it while the interpreter
    # blocks on the finished update instead of idling behind update_net's host sync (--eager-logs: read them at once)
    lazy = bool(getattr(agent, "supports_lazy_logs", False)) and not opt.eager_logs
    pend = []

    def step():
        items = explore_env(env, HORIZON)
        out = pend.pop().result() if pend else None
        r = update_net(list(items), lazy=True) if lazy else update_net(list(items))
        if hasattr(r, "result"):
            pend.append(r)
            return out
        return r

    def flush():
        return pend.pop().result() if pend else None

    # The GAE size sweep (part of the line: `roofline_gae.sweep`) runs BEFORE the warm-up since round 6 (`--gae-sweep-after` restores the old
    # order): it is ~0.2 s of HBM-bound GPU work, and a process that starts its timed region 11 ms after its first launch measures the
    # clocks' ramp, not the loop -- round 5's driver lines read 2.25 ms per step in the primary region next to 2.09-2.10 in the five regions
    # that followed it in the same process (`extra.repeated_regions_ms_per_step`; update_net fell from 2.15 to 1.90 ms ACROSS the region).
    # The timed region itself is unchanged: W warm-up steps, barrier + synchronize, exactly K steps, barrier + synchronize.
    sweep_early = None
    if not opt.no_gae_sweep and opt.config == "c4" and not opt.gae_sweep_after:
        log("GAE size sweep (before the warm-up)")
        sweep_early = gae_sweep(ops, dev)
        th.cuda.synchronize()
        th.cuda.empty_cache()
    # Round 6: nothing slow sits between the warm-up and the timed region any more.  The interpreter's full collection (gc.collect: 40-65 ms
    # with the GPU idle) and the 200 empty launches of the null-bracket measurement used to run AFTER the warm-up: the chip dropped its clocks
    # during that pause and the timed region measured them coming back (update_net 2.14 -> 1.88 ms across the 20 steps of the primary
    # region, 2.07 ms in every region after it: profiles/r06_bench_region_order.txt).  Now: collect + freeze and the null bracket first,
    # then the W warm-up steps, then -- with the collector switched off for the region, as timeit does -- barrier + synchronize and the K
    # timed steps.  `--gc-after-warmup` restores the old order.
    import gc
    if not opt.gc_after_warmup:
        quiet_gc()
        null_bracket_us = _hip.k6_null_bracket_us(200)   # what an event bracket adds to its content on this box (empty launch)
    log(f"rank {rank}/{world}: agent + env ready, warm-up x{opt.warmup}")
    for _ in range(opt.warmup):
        step()
    flush()
    th.cuda.synchronize()
    if opt.gc_after_warmup:
        quiet_gc()
        null_bracket_us = _hip.k6_null_bracket_us(200)
    else:
        gc.disable()
    log("timed region")
    t_gae.enabled = t_explore.enabled = t_update.enabled = True
    # every n-th K6 launch is timed twice: a HIP-event bracket on its stream and the kernel's own span on the device clock
    _hip.k6_timing_enable(opt.k6_sample)
    _hip.kernel_span_enable(8)             # every 8th launch of the loop's other kernels leaves its own device-clock span (no brackets): GAE scan, slab reduction, clip + Adam
    parallel.barrier()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        objs = step()
    objs = flush() or objs
    th.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0                  # this rank's own time to its last kernel (before the closing barrier)
    gc.enable()
    parallel.barrier()
    elapsed = parallel.all_reduce_max_float(time.perf_counter() - t0, device=dev)
    rank_ms_max = parallel.all_reduce_max_float(own_elapsed, device=dev) / opt.steps * 1e3
    rank_ms_min = -parallel.all_reduce_max_float(-own_elapsed, device=dev) / opt.steps * 1e3
    t_gae.enabled = t_explore.enabled = t_update.enabled = False
    _hip.k6_timing_enable(False)
    k6_event_seconds, k6_span_seconds, k6_launches = _hip.k6_timing_read2()
    span_gae_us, span_gae_n = _hip.kernel_span_read(_hip.SPAN_GAE)
    span_red_us, span_red_n = _hip.kernel_span_read(_hip.SPAN_SLAB_REDUCE)
    span_adam_us, span_adam_n = _hip.kernel_span_read(_hip.SPAN_CLIP_ADAM)
    _hip.kernel_span_enable(False)
    # every K6 launch of the region left its own span / clock / phase stamps; the launches WITHOUT an event bracket are the kernel as
    # the loop runs it (a bracket perturbs what it brackets: the bracketed group is reported next to it)
    k6_clocks, k6_clocks_br = _hip.k6_timing_clocks(False), _hip.k6_timing_clocks(True)
    k6_wgs = _hip.k6_wg_summary(_hip.k6_timing_last_records(False))      # where / when the last sampled launch's workgroups ran
    # (launch number since enable, span) of every sampled launch, bracketed or not (a bracket does not change the span inside it)
    k6_spans = _hip.k6_timing_spans(False) + _hip.k6_timing_spans(True)
    # 2: the update loop ran the two networks as two chains of half-chip launches on two streams (include/erl_hip.h erl_ppo_update_chains):
    # a minibatch is then TWO minibatch-kernel launches (actor, critic: 128 workgroups each at config 4), in flight side by side
    k6_chains = max(1, _hip.ppo_update_chains())
    smi = smi_snapshot() if rank == 0 and not opt.no_smi else None

    log(f"timed region done: {elapsed:.3f}s for {opt.steps} steps")
    # box variance made visible next to the driver's single sample: the same region repeated (not part of `value`)
    repeats = []
    for _ in range(opt.repeats):
        parallel.barrier()
        th.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(opt.steps):
            step()
        flush()
        th.cuda.synchronize()
        parallel.barrier()
        repeats.append(parallel.all_reduce_max_float(time.perf_counter() - t1, device=dev) / opt.steps * 1e3)
    # the same region on the fp32-MFMA minibatch kernel (when the default is the split-arithmetic one), for the record
    f32_region = None
    k6_arith = ops.ppo_arith_in_use(STATE_DIM, NET_DIMS[0], NET_DIMS[1], ACTION_DIM) if len(NET_DIMS) == 2 else "f32"
    wide = bool(getattr(agent, "_wide", False))             # net_dims (256, h2): one kernel, split arithmetic only
    if k6_arith == "split" and opt.repeats and not wide:
        prev_arith, agent.ppo_arith = agent.ppo_arith, "f32"      # (the agent applies its ppo_arith at every update_net)
        for _ in range(2):
            step()
        flush()
        parallel.barrier()
        th.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(opt.steps):
            objs_f32 = step()
        objs_f32 = flush() or objs_f32
        th.cuda.synchronize()
        parallel.barrier()
        el = parallel.all_reduce_max_float(time.perf_counter() - t1, device=dev)
        agent.ppo_arith = prev_arith
        ops.ppo_set_arith(prev_arith)
        f32_region = {"ms_per_step": round(el / opt.steps * 1e3, 3), "value": round(world * N_ENVS * HORIZON * opt.steps / el, 1),
                      "objectives_last": [round(float(x), 6) for x in objs_f32],
                      "note": "one more region of the same steps with erl_ppo_set_arith(f32): the fp32-MFMA minibatch kernel (not part of `value`; "
                              "its objectives belong to later iterations of the same training run than `objectives_last`)"}
    allreduce = None
    if world > 1 or parallel.force_dp():   # the exchange step on its own, through the route update_net uses (every rank takes part)
        comm = parallel.gradient_comm(agent._stride)
        buf = th.zeros(agent._stride, dtype=th.float32, device=dev)
        red = (lambda: comm.all_reduce_sum(buf)) if comm is not None else (lambda: parallel.all_reduce_sum(buf))
        for _ in range(20):
            red()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        th.cuda.synchronize()
        parallel.barrier()
        e0.record()
        for _ in range(200):
            red()
        e1.record()
        th.cuda.synchronize()
        us = parallel.all_reduce_max_float(e0.elapsed_time(e1) * 5.0, device=dev)      # ms / 200 calls -> us per call
        rep = parallel.route_report()      # what the start-up self-test measured and chose (both library routes)
        allreduce = {"selected": rep.get("selected"), "mode": rep.get("mode"), "rccl_us": rep.get("rccl_us"), "p2p_us": rep.get("p2p_us"),
                     "rccl_selftest": rep.get("rccl_selftest"), "p2p_selftest": rep.get("p2p_selftest"), "p2p_probe": rep.get("p2p_probe"),
                     "ranks_seen_by_rccl": rep.get("ranks_seen_by_rccl"), "bytes": int(buf.numel() * 4),
                     "selected_us_per_call_after_run": round(us, 2), "calls_per_step": UPDATE_TIMES,
                     "stats_exchange": "same route (fp64, 8 sums per iteration)" if comm is not None else "torch.distributed"}
    if rank != 0:
        return
    env_steps = world * N_ENVS * HORIZON * opt.steps
    flops = ppo_flops_per_sample(STATE_DIM, *NET_DIMS, ACTION_DIM) * BATCH
    n_k6 = k6_launches
    k6_event_s = k6_event_seconds / n_k6 if n_k6 else float("nan")
    k6_span_br_s = k6_span_seconds / n_k6 if n_k6 and k6_span_seconds > 0 else float("nan")      # spans of the BRACKETED launches
    k6_span_s = k6_clocks["span_us"] * 1e-6 if k6_clocks["launches"] else k6_span_br_s           # ... of the launches without a bracket
    # the kernel's duration: its own first-workgroup-in to last-workgroup-out span on the device's constant-rate clock (no dispatch or
    # completion overhead of a bracket in it; agrees with rocprofv3's kernel duration -- `kernel_us_rocprof`); fallback: the event
    # bracket minus the bracket of an empty launch
    ppo_s = k6_span_s if k6_span_s == k6_span_s else max(k6_event_s - null_bracket_us * 1e-6, 1e-9)
    # where in the update loop a sampled launch sat: the loop's FIRST launch follows the rollout and the GAE scan and finds the instruction
    # caches (and the XCDs' L2s) without the kernel's code -- +3 us on most boxes of the pool, +50 us on the ones with slow instruction
    # fetch (DESIGN.md section 4 "Instruction fetch and the workgroup map"; profiles/HISTORY.md "K6 in round 5").  avg_launch_us weights the two groups as the loop does (1 : update_times - 1), whatever the
    # sampling period made of them.
    k6_lpl = k6_chains * UPDATE_TIMES                 # minibatch-kernel launches per update loop
    k6_first = [us for k, us in k6_spans if k % k6_lpl < k6_chains]
    k6_rest = [us for k, us in k6_spans if k % k6_lpl >= k6_chains]
    k6_by_position = None
    if k6_rest:
        rest_us = sum(k6_rest) / len(k6_rest)
        first_us = sum(k6_first) / len(k6_first) if k6_first else rest_us
        k6_by_position = {"first_launch_of_a_loop_us": round(first_us, 2) if k6_first else None, "first_launches_sampled": len(k6_first),
                          "other_launches_us": round(rest_us, 2), "other_launches_sampled": len(k6_rest),
                          "unweighted_mean_us": round(ppo_s * 1e6, 2)}
        if k6_chains == 2:                              # (the loop enqueues actor, critic, actor, ...: the launch number's parity is the network)
            for net, name in ((0, "actor"), (1, "critic")):
                v = [us for k, us in k6_spans if k % 2 == net and k % k6_lpl >= k6_chains]
                k6_by_position[f"{name}_launches_us"] = round(sum(v) / len(v), 2) if v else None
        if UPDATE_TIMES > 1:
            ppo_s = (first_us + (UPDATE_TIMES - 1) * rest_us) / UPDATE_TIMES * 1e-6
    # which K6 kernel erl_ppo_step_f32 dispatches to: the one-wave-per-SIMD form for h1, h2 in {64, 128}, S <= 64, A <= 8
    k6_kernel = "ppo_step_w4_kernel" if (len(NET_DIMS) == 2 and all(d in (64, 128) for d in NET_DIMS) and STATE_DIM <= 64
                                         and ACTION_DIM <= 8) else "ppo_step2_kernel"
    if k6_arith == "split":
        k6_kernel = "ppo_step_wd_kernel" if wide else "ppo_step_s3_kernel"
    # the fp32-equivalent ceiling of the pipe the kernel runs on: the split-arithmetic kernel issues SPLIT_TERMS bf16 MFMA flops per
    # algorithmic flop on the bf16 matrix pipe; the fp32 kernel runs on the fp32 MFMA
    k6_peak = MFMA_BF16_PEAK_TFLOPS / SPLIT_TERMS if k6_arith == "split" else MFMA_F32_PEAK_TFLOPS
    gae_s = t_gae.mean_seconds()                      # HIP-event bracket around the ops.gae_scan call (launch + bracket + kernel)
    gae_k_s = span_gae_us * 1e-6 if span_gae_us else gae_s     # ... the kernel's own span
    k6_traffic, k6_traffic_src = (pmc_traffic(k6_kernel) if opt.config == "c4" else
                                  pmc_traffic(k6_kernel, WIDE_PMC_FILE) if opt.config == "cw" else (None, None))
    k6_rocprof_us, k6_rocprof_src = rocprof_kernel_us(k6_kernel) if opt.config == "c4" else (None, None)
    k6_mhz = float(k6_clocks.get("shader_mhz") or 0.0)
    # live kernel time against the committed rocprofv3 average of the same command on the same sources ("the two must agree"): a
    # ratio beyond 10 % means THIS box runs the kernel at another speed than the box the profile came from -- said out loud, with
    # the clock and the per-phase cycles next to it, instead of two disagreeing numbers in one line
    box_ratio = round(ppo_s * 1e6 / k6_rocprof_us, 3) if k6_rocprof_us else None
    phase_ref = None
    try:
        phase_ref = json.load(open(os.path.join(ROOT, KTIME_FILE))).get("phase_cycles_reference")
    except Exception:
        pass
    if box_ratio is not None and abs(box_ratio - 1.0) > 0.10:
        log(f"WARNING: the minibatch kernel runs {ppo_s * 1e6:.1f} us here against {k6_rocprof_us:.1f} us in {KTIME_FILE} (x{box_ratio}); shader clock "
            f"in the kernel {k6_mhz:.0f} MHz, phase cycles {k6_clocks['phase_cycles']} (reference: {phase_ref})")
    # the parts of a step against the step: explore_env + update_net brackets (each carries one bracket overhead) must fit into
    # ms_per_step, and the K6 launches must fit into update_net
    explore_ms, update_ms = t_explore.mean_seconds() * 1e3, t_update.mean_seconds() * 1e3
    k6_ms = UPDATE_TIMES * ppo_s * 1e3
    step_ms = elapsed / opt.steps * 1e3
    breakdown = {"update_loop_chains": k6_chains,     # 2: actor and critic chains side by side -- k6_ms / slab_reduce_us / clip_adam_us are ONE chain's launches
                 "explore_env_ms": round(explore_ms, 4), "update_net_ms": round(update_ms, 4),
                 "k6_ms": round(k6_ms, 4), "update_net_minus_k6_ms": round(update_ms - k6_ms, 4),
                 "per_minibatch_rest_us": round((update_ms - k6_ms) / UPDATE_TIMES * 1e3, 2),
                 "host_and_gaps_ms": round(step_ms - explore_ms - update_ms, 4),
                 # the tail's kernels by their own device-clock spans (first workgroup in to last workgroup out, every launch of the region)
                 "slab_reduce_us": round(span_red_us, 2) if span_red_us else None, "clip_adam_us": round(span_adam_us, 2) if span_adam_us else None,
                 "boundaries_and_rest_per_minibatch_us": (round((update_ms - k6_ms) / UPDATE_TIMES * 1e3 - span_red_us - span_adam_us, 2)
                                                          if span_red_us and span_adam_us else None),
                 # what the update loop leaves for a minibatch-kernel launch AND its boundaries: update_net / update_times - the tail's spans.
                 # avg_launch_us above it (boundaries_and_rest < 0) means the SAMPLED launches ran longer than the loop's mean launch on this
                 # box (seen on the pool's slow boxes, where a launch that leaves records is ~9 us slower than one that does not)
                 "k6_us_upper_bound_by_difference": (round(update_ms / UPDATE_TIMES * 1e3 - span_red_us - span_adam_us, 2)
                                                     if span_red_us and span_adam_us else None),
                 "explore_env_ms_each": t_explore.each_ms(), "update_net_ms_each": t_update.each_ms(),
                 "consistent": bool(k6_ms <= update_ms and explore_ms + update_ms <= step_ms * 1.01),
                 "note": "HIP-event brackets around agent.explore_env / agent.update_net (means over the timed region); k6_ms = update_times x "
                         "roofline.avg_launch_us; per_minibatch_rest_us = slab reduction + clip/Adam + launch boundaries (+ GAE, statistics, "
                         "index draw, log fold, amortised over the minibatches); consistent = the parts fit into the step"}
    if not breakdown["consistent"]:
        log(f"WARNING: timing parts do not fit into the step: {breakdown}")
    if (breakdown["boundaries_and_rest_per_minibatch_us"] or 0.0) < 0.0:
        log(f"WARNING: the sampled minibatch-kernel launches ({ppo_s * 1e6:.1f} us) ran longer than the update loop leaves for a launch "
            f"({breakdown['k6_us_upper_bound_by_difference']} us incl. its boundaries): on this box the sampled launches are not representative")
    wg_info = _hip.ppo_wg_map_info(wide=wide)
    if wg_info.get("us_map0") and wg_info.get("us_map2") and wg_info["us_map0"] > 1.15 * wg_info["us_map2"]:
        log(f"NOTE: this box's instruction caches miss slowly (minibatch kernel back to back: {wg_info['us_map0']} us with both networks' code paths behind "
            f"every instruction cache, {wg_info['us_map2']} us with one): the library chose workgroup map {wg_info['map']} and the code touch (DESIGN.md section 4, 'Instruction fetch and the workgroup map')")
    line = {
        "metric": cfg["metric"], "value": round(env_steps / elapsed, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": round(elapsed / opt.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["workload"], "name": opt.config,
                   "envs_per_gpu": N_ENVS, "horizon": HORIZON, "batch": BATCH, "update_times": UPDATE_TIMES,
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "last_state": "private copy (reference behaviour)" if agent.snapshot_last_state else "aliases the env's live state buffer",
                   "interpreter": ("gc.collect() + gc.freeze() after the warm-up (as elegantrl_amd.train.run does)" if opt.gc_after_warmup else
                                   "gc.collect() + gc.freeze() before the warm-up, collector off for the timed region (nothing slow between warm-up and region)"),
                   "logs": ("update_net's three logged objectives are read one rollout late (update_net(lazy=True), as elegantrl_amd.train.run does): "
                            "no host sync between an update and the next rollout's launch; every kernel of the K steps and the last read are inside the timed region"
                            if lazy else "read at once (one host sync per iteration)"),
                   "k6_arith": ("split: fp32 operands as three bf16 parts on the bf16 matrix pipe, fp32 accumulate (as close to fp64 as the "
                                "fp32 MFMA: tests/test_kernels_gpu.py::test_ppo_step_split_arith)" if k6_arith == "split" else "f32 MFMA")},
        "roofline": {"kernel": k6_kernel, "bound": "mfma", "achieved": round(flops / k6_chains / ppo_s / 1e12, 2),
                     "peak": round(k6_peak / k6_chains, 1), "unit": "TFLOP/s", "frac": round(flops / ppo_s / 1e12 / k6_peak, 4),
                     "launch_form": ("one launch per minibatch over both networks (the whole chip)" if k6_chains == 1 else
                                     "two-chain update loop: a minibatch is TWO launches of this kernel, the actor's and the critic's, on two streams; each "
                                     "holds half of the chip's CUs (128 of 256 workgroup slots at config 4), so `achieved` = one launch's algorithmic "
                                     "flops (the mean of the two networks') / one launch's mean duration and `peak` = the chip's peak x 1/2; both launches "
                                     "of a minibatch are in flight side by side"),
                     "arith": ("fp32-equivalent: operands split into three bf16 parts, six partial products per product on v_mfma_f32_32x32x16_bf16, "
                               "fp32 accumulation; `achieved` counts ALGORITHMIC flops, `peak` = bf16 dense MFMA peak / 6"
                               if k6_arith == "split" else "fp32 operands on v_mfma_f32_32x32x2_f32"),
                     "frac_of_fp32_mfma_peak": round(flops / ppo_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": k6_traffic, "traffic_source": k6_traffic_src, "flops_per_launch": flops // k6_chains,
                     "avg_launch_us": round(ppo_s * 1e6, 2), "launches_timed": k6_clocks["launches"] or n_k6,
                     "by_position_in_the_update_loop": k6_by_position,
                     "timer": ("kernel span on the device clock: first workgroup in to last workgroup out (wall_clock64 in the kernel, one record per "
                               f"workgroup), over the sampled launches of the timed region that carry NO event bracket (1 in {opt.k6_sample}), the loop's "
                               "first launch weighted 1 : update_times - 1 against the others"
                               if k6_clocks["launches"] else
                               "kernel span on the device clock, bracketed launches" if k6_span_s == k6_span_s else
                               "HIP-event bracket minus empty-launch bracket"),
                     # every opt.k6_sample-th launch also sits inside a HIP-event bracket: the bracket's own time, the SAME launches' in-kernel
                     # span, and how much longer a bracketed launch runs than its unbracketed neighbours on this box (round 5: a bracket
                     # does not perturb the kernel inside it -- the two groups agree to < 1 us on every box sampled)
                     "bracketed": {"launches": n_k6, "event_bracket_us": round(k6_event_s * 1e6, 2),
                                   "span_us": round(k6_span_br_s * 1e6, 2) if k6_span_br_s == k6_span_br_s else None,
                                   "span_minus_unbracketed_us": round((k6_span_br_s - k6_span_s) * 1e6, 2) if k6_span_br_s == k6_span_br_s else None,
                                   "shader_mhz": round(k6_clocks_br["shader_mhz"], 1) if k6_clocks_br["shader_mhz"] else None,
                                   "phase_cycles": {k: round(v) for k, v in k6_clocks_br["phase_cycles"].items()} or None},
                     "event_bracket_us": round(k6_event_s * 1e6, 2), "event_bracket_null_us": round(null_bracket_us, 2),
                     # where the kernel's workgroups run (include/erl_hip.h erl_ppo_wg_map_info): the device's first full-chip launch
                     # measured map 0 against map 2 (us_map0 / us_map2, back to back) and kept one -- map 2 on the boxes where two code paths
                     # per instruction cache cost 7-9 us per launch (DESIGN.md section 4 "Instruction fetch and the workgroup map"; profiles/HISTORY.md "K6 in round 5"), map 0 elsewhere
                     "workgroup_map": wg_info,
                     # what that measurement says about the box: two ~58 KB code paths per instruction cache (map 0) cost > 15 % against one
                     # (map 2) only where the instruction caches' miss path is slow (~1 box in 4-20 of the pool)
                     "instruction_fetch": (None if not (wg_info.get("us_map0") and wg_info.get("us_map2")) else
                                           "slow" if wg_info["us_map0"] > 1.15 * wg_info["us_map2"] else "normal"),
                     "kernel_us_rocprof": k6_rocprof_us, "kernel_us_rocprof_source": k6_rocprof_src,
                     # this box against the box the committed rocprofv3 summary was collected on (same sources): avg_launch_us / kernel_us_rocprof
                     "box_ratio": box_ratio,
                     # the clock the sampled launches actually ran at (shader cycles per constant-rate tick inside the kernel) and the
                     # fraction against the peak scaled to THAT clock: what the kernel does with the cycles this box gave it
                     "shader_mhz": round(k6_mhz, 1) if k6_mhz else None,
                     "frac_at_measured_clock": round(flops / ppo_s / 1e12 / (k6_peak * k6_mhz / SHADER_PEAK_MHZ), 4) if k6_mhz else None,
                     "phase_cycles": {k: round(v) for k, v in k6_clocks["phase_cycles"].items()} or None,
                     "phase_cycles_reference": phase_ref},
        "clocks": {"shader_mhz_in_k6": round(k6_mhz, 1) if k6_mhz else None, "peak_quoted_at_mhz": SHADER_PEAK_MHZ,
                   "k6_workgroup_us": round(k6_clocks["workgroup_us"], 2), "phase_workgroups": k6_clocks["phase_workgroups"],
                   "k6_workgroups_last_sampled_launch": k6_wgs,
                   "smi": smi,
                   "note": "shader_mhz_in_k6 = sum over workgroups of (s_memtime exit - entry) / (constant-rate clock exit - entry) x its rate, on the "
                           "sampled minibatch-kernel launches of the timed region; smi = rocm-smi / amd-smi right after the region"},
        "breakdown": breakdown,
        "roofline_gae": ({"kernel": f"{'gae_exact_kernel' if HORIZON < 64 else 'gae_tall_kernel' if HORIZON <= 256 else 'gae_lookback_kernel'} (in-loop {HORIZON}x{N_ENVS})",
                          "bound": "hbm",
                          "achieved": round(18.0 * HORIZON * N_ENVS / gae_k_s / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                          "frac": round(18.0 * HORIZON * N_ENVS / gae_k_s / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                          "bytes_per_launch": 18 * HORIZON * N_ENVS, "avg_launch_us": round(gae_k_s * 1e6, 2), "launches_timed": span_gae_n,
                          "timer": "the kernel's own span on the device clock, every in-loop launch" if span_gae_us else "HIP-event bracket around the call",
                          "call_bracket_us": round(gae_s * 1e6, 2)} if t_gae.pairs else
                         {"kernel": f"none in the loop: at {HORIZON}x{N_ENVS} get_advantages and its statistics run in the persistent rollout's "
                                    "epilogue (csrc/rollout_fused.hip; ERL_FUSED_GAE=0 restores the scan launches); the scan kernel's HBM figures: `sweep`",
                          "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None,
                          "bytes_per_launch": 18 * HORIZON * N_ENVS, "avg_launch_us": None}),
        "objectives_last": [round(float(x), 6) for x in objs],
    }
    if opt.config == "cd":
        # at this shape the rollout (2048 steps x 4096 envs in ONE launch) is the dominant kernel: it becomes `roofline`, the minibatch
        # kernel (2 workgroups per launch at B = 128: latency, not throughput) moves to `roofline_k6`
        line["roofline_k6"] = line["roofline"]
        rflops = rollout_flops_per_env_step(STATE_DIM, *NET_DIMS, ACTION_DIM) * N_ENVS * HORIZON
        rs = explore_ms * 1e-3
        peak = MFMA_BF16_PEAK_TFLOPS / SPLIT_TERMS
        line["roofline"] = {"kernel": "rollout_fused_kernel", "bound": "mfma", "achieved": round(rflops / rs / 1e12, 2), "peak": round(peak, 1),
                            "unit": "TFLOP/s", "frac": round(rflops / rs / 1e12 / peak, 4), "traffic": None, "flops_per_launch": rflops,
                            "avg_launch_us": round(rs * 1e6, 1), "launches_timed": len(t_explore.pairs),
                            "us_per_env_step_row": round(rs * 1e6 / HORIZON, 3),
                            "arith": "hidden layers: fp32 operands as three bf16 parts, six partial products on v_mfma_f32_16x16x32_bf16 (csrc/rollout_bf16.h); "
                                     "output layers and the env's matrices on the fp32 MFMA; `peak` = bf16 dense MFMA peak / 6",
                            "timer": "HIP-event bracket around agent.explore_env (one launch + its allocations)",
                            "note": "a 16-env tile per workgroup walks all H steps: 256 workgroups x 8 waves, a step is a chain of dependent small "
                                    "layers (latency-bound by construction: DESIGN.md, Persistent rollout)"}
    if allreduce is not None:
        line["allreduce"] = allreduce
    if repeats:
        srt = sorted(repeats)
        line["extra"] = {"repeated_regions_ms_per_step": [round(x, 3) for x in repeats], "min": round(srt[0], 3),
                         "median": round(srt[len(srt) // 2], 3), "max": round(srt[-1], 3), "primary": round(elapsed / opt.steps * 1e3, 3),
                         "note": f"{len(repeats)} more timed regions of {opt.steps} steps each after the primary one (not part of `value`)"}
    if f32_region is not None:
        line.setdefault("extra", {})["k6_arith_f32"] = f32_region
    if world > 1:
        line.setdefault("extra", {})["per_rank_ms_per_step"] = {"min": round(rank_ms_min, 3), "max": round(rank_ms_max, 3),
                                                                 "note": "every rank's own time to its last kernel, before the closing barrier"}
    if not opt.no_gae_sweep and opt.config == "c4":
        if sweep_early is None:
            log("GAE size sweep")
        sweep = sweep_early if sweep_early is not None else gae_sweep(ops, dev)
        line["roofline_gae"]["sweep_ran"] = "before the warm-up" if sweep_early is not None else "after the timed region"
        line["roofline_gae"]["sweep"] = sweep
        big = next(x for x in sweep if (x["H"], x["N"]) == (2048, 4096))
        gae_tr, gae_tr_src = pmc_traffic("gae_lookback_kernel")
        line["roofline_gae"]["at_2048x4096"] = {"kernel": "gae_lookback_kernel (library-owned granule table, no memset)", "achieved": big["GBps"],
                                                "frac": big["frac"], "us": big["us"], "bytes_per_launch": big["bytes"],
                                                "traffic": gae_tr, "traffic_source": gae_tr_src}
    if world == 1 and not opt.no_cpu_baseline:
        log(f"cpu baseline ({usable_cores()} usable cores of {os.cpu_count()})")
        # (cd: one CPU iteration is 2048 x 4096 env steps + a value pre-pass over 8.4 M rows, ~20 s on 16 cores: warm-up + 1)
        line["cpu_baseline"] = cpu_baseline_subprocess(opt.cpu_iters if opt.config == "c4" else 1 if opt.config in ("cd", "c1") else max(2, opt.cpu_iters // 4),
                                                       opt.config, timeout_s=600 if opt.config == "cd" else 300)
    log("done")
    emit(line)


if __name__ == "__main__":
    main()
    from elegantrl_amd import parallel as _parallel
    _parallel.shutdown()      # every rank: barrier -> RCCL communicator -> process group (rank 0 has printed its line by now)
