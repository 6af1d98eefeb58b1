"""bench.py itself: the contract line at N = 1 (every configuration) and the N > 1 code path (two ranks sharing the GPU over
gloo: the barrier / max-over-ranks timing, the gradient exchange in the loop and the all-reduce micro-benchmark run for real)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def run(cmd, env=None):
    base = {k: v for k, v in os.environ.items() if k != "ERL_K6_ARITH"}       # the contract is about the library's default arithmetic
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(base, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("config", ["c4", "c5", "c2", "c3"])
def test_bench_line_contract(config):
    line = run([sys.executable, "bench.py", "--config", config, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-gae-sweep"])
    assert KEYS <= set(line) and line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    assert line["config"]["name"] == config and "workload" in line["config"] and line["vs_baseline"] is None
    r = line["roofline"]
    assert {"kernel", "bound", "achieved", "peak", "unit", "frac"} <= set(r) and 0 < r["frac"] < 1
    if config == "c4":
        # the default minibatch kernel of [128,128] nets: split arithmetic on the bf16 matrix pipe, priced against that pipe's
        # fp32-equivalent ceiling (bf16 dense peak / 6 partial products); the fp32-MFMA region rides along in `extra`
        assert line["metric"] == "env_steps_per_sec_ppo_4096envs_obs64" and r["kernel"] == "ppo_step_s3_kernel"
        assert abs(r["peak"] - 2500.0 / 6) < 0.1 and "split" in line["config"]["k6_arith"] and r["frac_of_fp32_mfma_peak"] > r["frac"]
        f32 = line["extra"]["k6_arith_f32"]
        assert f32["value"] > 0 and len(f32["objectives_last"]) == 3
        # HBM bytes per launch from the committed PMC passes, or None when they were collected on other kernel sources
        src = r["traffic_source"]
        assert (r["traffic"] is None) == bool(src["stale"]) and (r["traffic"] is None or r["traffic"] > 0)
        assert len(src["kernel_source_sha16"]) == 16
        e = line["extra"]
        assert len(e["repeated_regions_ms_per_step"]) == 5 and e["min"] <= e["median"] <= e["max"]


def test_bench_two_ranks_on_one_gpu_over_gloo():
    line = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29561", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-gae-sweep"],
               env={"ERL_DIST_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    a = line["allreduce"]
    assert a["bytes"] == 4 * ((25872 + 24961 + 4 + 31) // 32 * 32) and a["selected_us_per_call_after_run"] > 0 and a["calls_per_step"] == 40
    # two ranks on one GPU: RCCL cannot come up (gloo group), the self-tested peer-to-peer exchange must be the route
    assert a["mode"] == "auto" and a["rccl_selftest"] == "not tried" and a["p2p_selftest"] == "ok" and a["p2p_us"] > 0, a
    assert "peer-to-peer" in a["selected"], a
    assert "cpu_baseline" not in line


def _torchrun(n, port, extra, env):
    return run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-gae-sweep", "--no-smi", *extra], env=env)


def test_bench_eight_ranks_on_one_gpu_rehearsal():
    """`bench.py --gpus 8` as the driver launches it (torch.distributed.run, one rank per "GPU"), rehearsed with the eight ranks SHARING
    the one device over gloo on the small `cr` shape: the barrier / max-over-ranks contract, the route report (RCCL cannot come up on a gloo
    group: not tried; the out-of-process probe and the in-process self-test of the peer-to-peer exchange pass at world 8; it is selected),
    the exchange inside the loop (4 minibatches per step), every rank's own time.  No scaling number comes out of this: one GPU."""
    line = _torchrun(8, 29571, ["--config", "cr"], {"ERL_DIST_BACKEND": "gloo"})
    assert line["n_gpus"] == 8 and line["config"]["parallelism"] == "dp8" and line["scaling"] == "weak" and line["config"]["name"] == "cr"
    assert line["value"] > 0 and abs(line["value"] - 8 * 512 * 16 * 2 / (line["ms_per_step"] * 2e-3)) / line["value"] < 0.02
    a = line["allreduce"]
    assert a["mode"] == "auto" and a["rccl_selftest"] == "not tried" and a["ranks_seen_by_rccl"] is None, a
    assert a["p2p_probe"] == "ok" and a["p2p_selftest"] == "ok" and a["p2p_us"] > 0 and "peer-to-peer" in a["selected"], a
    assert a["calls_per_step"] == 4 and a["selected_us_per_call_after_run"] > 0 and a["stats_exchange"].startswith("same route")
    pr = line["extra"]["per_rank_ms_per_step"]
    assert 0 < pr["min"] <= pr["max"] <= line["ms_per_step"] * 1.05
    assert "cpu_baseline" not in line and len(line["objectives_last"]) == 3 and all(x == x for x in line["objectives_last"])


def test_bench_falls_back_when_the_p2p_probe_child_dies():
    """the fallback ladder: the probe's child process is made to die (ERL_P2P_PROBE_FAIL: what a faulting peer mapping does to it) -- on
    a gloo group RCCL is not available either, so every rank must agree on torch.distributed, say why, and finish the run"""
    line = _torchrun(4, 29573, ["--config", "cr"], {"ERL_DIST_BACKEND": "gloo", "ERL_P2P_PROBE_FAIL": "7"})
    a = line["allreduce"]
    assert a["p2p_probe"] != "ok" and ("exited with 7" in a["p2p_probe"] or "another rank" in a["p2p_probe"]), a
    assert a["p2p_selftest"].startswith("not run") and a["rccl_selftest"] == "not tried" and a["selected"] == "torch.distributed", a
    assert line["n_gpus"] == 4 and line["value"] > 0 and a["stats_exchange"] == "torch.distributed"


def test_bench_gpus_2_starts_its_own_ranks():
    """plain `python bench.py --gpus 2` -- no torchrun around it, no WORLD_SIZE in the environment: bench.py starts the two ranks itself
    (`self_launch`), rank 0's single line comes through with n_gpus = 2, the exchange route it chose, what RCCL saw, every rank's own time"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "cr", "--steps", "2", "--warmup", "1", "--no-gae-sweep", "--no-smi"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(env, ERL_DIST_BACKEND="gloo"))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["name"] == "cr" and line["value"] > 0
    a = line["allreduce"]
    assert "ranks_seen_by_rccl" in a and a["selected"] and a["calls_per_step"] == 4
    pr = line["extra"]["per_rank_ms_per_step"]
    assert 0 < pr["min"] <= pr["max"]
