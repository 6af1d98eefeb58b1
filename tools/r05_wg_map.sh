#!/bin/bash
# round 5: the minibatch kernel's workgroup map (ERL_K6_WG_MAP, ppo_step.h k6_wg_map) on ONE box: in-process A/B of the kernel inside the
# update loop, the config-4 bench line under each map (alternating processes), and the parity tests under map 1.
TAG=${1:-w}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_wgmap_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
tools/bin/clock_probe > $O/clock_probe.json 2> $O/clock_probe.err
python tools/k6_wg_map_ab.py 4 > $O/ab.jsonl 2> $O/ab.err
for rep in 0 1; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/c4_auto_$rep.json 2> $O/c4_auto_$rep.err
  for m in 0 1 2; do
    ERL_K6_WG_MAP=$m python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/c4_map${m}_$rep.json 2> /dev/null
  done
done
if [ -z "$2" ]; then
  for m in 1 2; do ERL_K6_WG_MAP=$m timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py tests/test_oracle_golden.py tests/test_ppo_wide_gpu.py -m gpu -q -x > $O/pytest_map$m.log 2>&1; echo "map$m: $(tail -1 $O/pytest_map$m.log)"; done
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -s -k "workgroup_map" > $O/pytest_auto.log 2>&1; echo "auto: $(tail -1 $O/pytest_auto.log)"; grep "workgroup map on this box" $O/pytest_auto.log
fi
python - <<PY
import json, glob
try:
    cp = json.load(open("$O/clock_probe.json")); print("code_walk", cp.get("code_walk_56KB")); print("k6like", cp.get("k6_like_forward_mix"))
except Exception as e:
    print("clock_probe:", e)
for ln in open("$O/ab.jsonl"):
    r = json.loads(ln); print("ab map", r["wg_map"], "loop_ms", r["loop_ms_40_minibatches"], "span", r.get("us_span"), "wg", r.get("workgroup_us"), r.get("dur_us"), "mhz", r.get("shader_mhz"), "sum", r["params_checksum"], r.get("dur_us_mean_by_xcc"))
for f in sorted(glob.glob("$O/c4_*.json")):
    d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
    print(f.split('/')[-1], r.get("workgroup_map"), d["breakdown"].get("k6_us_upper_bound_by_difference"), d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "mhz", r["shader_mhz"], "reduce", b["slab_reduce_us"], "adam", b["clip_adam_us"])
PY
