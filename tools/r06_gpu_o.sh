#!/bin/bash
# K6 A/B of one compile-time switch: the named variant library against main, alternating (tests on main first)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_o; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py -m gpu -q -x -k "ppo or agent or golden or step or update" > $O/pytest_main.log 2>&1; echo "pytest main rc=$?" >> $O/pytest_main.log
tail -3 $O/pytest_main.log
for cfg in c4 c2; do
  for rep in 0 1 2; do
    for a in d3p0 main; do
      lib=$L/liberl_hip.so; [ $a != main ] && lib=$L/liberl_hip_$a.so
      ERL_HIP_LIB=$lib timeout 300 python bench.py --config $cfg --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/${cfg}_${a}_$rep.json 2> $O/${cfg}_${a}_$rep.err
    done
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c*_*_?.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "update_ms", b["update_net_ms"], "phases", r.get("phase_cycles"))
    except Exception as e:
        print(f, "FAILED", e)
PY
