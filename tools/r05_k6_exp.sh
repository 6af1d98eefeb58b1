#!/bin/bash
# round 5, K6 experiments on ONE box, alternating processes: -DERL_K6_EXP bits -- 1: the last dW2 pass leaves by plain stores, 2: all of dW2,
# 4: the critic's workgroups pull the next minibatch's rows towards their XCD's L2 -- against the default build.
TAG=${1:-x}; shift
VARS=${@:-main e1 e2 e4 e5}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_k6exp_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
for rep in 0 1; do
  for v in $VARS; do
    lib=$L/liberl_hip.so; [ $v != main ] && lib=$L/liberl_hip_$v.so
    [ -f $lib ] || continue
    ERL_HIP_LIB=$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/${v}_$rep.json 2> /dev/null
  done
done
for v in $VARS; do
  [ $v = main ] && continue
  ERL_HIP_LIB=$L/liberl_hip_$v.so timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py -m gpu -q -x -k "golden or update_loop or c4_iteration" > $O/pytest_$v.log 2>&1; echo "$v: $(tail -1 $O/pytest_$v.log)"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
    print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "mhz", r["shader_mhz"], "reduce", b["slab_reduce_us"], "adam", b["clip_adam_us"],
          "phases", [r["phase_cycles"][k] for k in ("prologue", "layer1_forward", "dW2_logs_drain")])
PY
