#!/bin/bash
# round 6, second session: the minibatch kernel's weight-gradient phases in the order dW1, dW2, dW3 (the 64 KB store first) against the
# order of rounds 3-5 (dW1, dW3, dW2), alternating processes on ONE box; the kernel tests on the variant; phase profiles of both.
#     gpurun -- bash tools/r06_k6_dwo.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r06_dwo; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
ERL_HIP_LIB=$L/liberl_hip_dwo.so timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py -m gpu -q -x -k "ppo or agent or golden or step" > $O/pytest_dwo.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dwo.log
tail -3 $O/pytest_dwo.log
for rep in 0 1; do
  for v in main dwo dwo4; do
    lib=$L/liberl_hip.so; [ $v != main ] && lib=$L/liberl_hip_$v.so
    [ -f $lib ] || continue
    ERL_HIP_LIB=$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/${v}_$rep.json 2> $O/${v}_$rep.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*_?.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "mhz", r.get("shader_mhz"), "fetch", r.get("instruction_fetch"), "reduce", b["slab_reduce_us"], "adam", b["clip_adam_us"],
              "phases", r.get("phase_cycles"))
    except Exception as e:
        print(f, "FAILED", e)
PY
for v in prof dwop; do
  ERL_HIP_PROF_LIB=$L/liberl_hip_$v.so K6_LOOP=1 python tools/ppo_phase_profile.py > $O/phase_$v.txt 2>&1
  grep -A16 "actor: total" $O/phase_$v.txt | cut -c1-120
done
