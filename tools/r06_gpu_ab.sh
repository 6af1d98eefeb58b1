#!/bin/bash
# generic A/B of variant libraries against main on config 4: VARIANTS="a b" bash tools/r06_gpu_ab.sh tag   (alternating processes, one box)
TAG=${1:-ab}; O=$GRAFT_REPO_ROOT/gpurun_out/r06_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
for rep in 0 1 2; do
  for a in $VARIANTS main; do
    lib=$L/liberl_hip.so; [ $a != main ] && lib=$L/liberl_hip_$a.so
    ERL_HIP_LIB=$lib timeout 300 python bench.py --config ${CFG:-c4} --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/c4_${a}_$rep.json 2> $O/c4_${a}_$rep.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c4_*_?.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "phases", r.get("phase_cycles"))
    except Exception as e:
        print(f, "FAILED", e)
PY
