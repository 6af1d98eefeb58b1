#!/bin/bash
# round 5: the code touch of the update loop's first minibatch-kernel launch (ppo_step.h k6_code_touch) on ONE box: config 4 with the touch
# forced off / forced on / left to the device's own classification, alternating processes; parity tests with the touch forced on.
TAG=${1:-t}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_touch_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
tools/bin/clock_probe > $O/clock_probe.json 2> $O/clock_probe.err
for rep in 0 1; do
  for t in 0 1 auto; do
    if [ $t = auto ]; then python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 --k6-sample 3 > $O/c4_touch${t}_$rep.json 2> /dev/null
    else ERL_K6_CODE_TOUCH=$t python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 --k6-sample 3 > $O/c4_touch${t}_$rep.json 2> /dev/null; fi
  done
done
if [ -z "$2" ]; then
  ERL_K6_CODE_TOUCH=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py tests/test_oracle_golden.py -m gpu -q -x > $O/pytest_touch1.log 2>&1; echo "touch1: $(tail -1 $O/pytest_touch1.log)"
fi
python - <<PY
import json, glob
cp = json.load(open("$O/clock_probe.json")); print("code_walk", {k: v["back_to_back_us"] for k, v in cp["code_walk_56KB"].items()}, "first", cp["code_walk_56KB"]["single"]["first_launch_span_us"])
for f in sorted(glob.glob("$O/c4_*.json")):
    d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]; p = r["by_position_in_the_update_loop"]
    print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "first", p["first_launch_of_a_loop_us"], p["first_launches_sampled"], "others", p["other_launches_us"], r["workgroup_map"], "update_ms", b["update_net_ms"])
PY
