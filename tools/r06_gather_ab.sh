#!/bin/bash
# round 6, last session: K6 prologue -- row-number shuffles first + unconditional row loads + the unmask compare deferred (ERL_K6_GATHER=1, main)
# against the earlier form (liberl_hip_g0.so: -DERL_K6_GATHER=0 in ppo_step_s3_pre.hip); tests first, then config 4 alternating, then the phase profile
#   gpurun -- bash tools/r06_gather_ab.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r06_gather; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py -m gpu -x -q -k "ppo or agent or step" 2>&1 | tail -3
for rep in 0 1 2; do
  for a in g0 main; do
    lib=$L/liberl_hip.so; [ $a != main ] && lib=$L/liberl_hip_$a.so
    ERL_HIP_LIB=$lib timeout 300 python bench.py --config c4 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/c4_${a}_$rep.json 2> $O/c4_${a}_$rep.err
  done
done
ERL_HIP_PROF_LIB=$L/liberl_hip_prof.so K6_LOOP=1 python tools/ppo_phase_profile.py > $O/k6_phase_c4.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c*_*_?.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "mhz", r.get("shader_mhz"), "phases", r.get("phase_cycles"))
    except Exception as e:
        print(f, "FAILED", e)
PY
grep -E "prologue|barrier0|total cycles" $O/k6_phase_c4.txt | head -8
