#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r06_sacx3; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
for rep in 0 1; do
  python bench.py --config c3 --no-cpu-baseline > $O/c3_wpe4_$rep.json 2> /dev/null
  ERL_HIP_LIB=$L/liberl_hip_wpe2.so python bench.py --config c3 --no-cpu-baseline > $O/c3_wpe2_$rep.json 2> /dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c3_*.json")):
    d = json.loads(open(f).readline()); r = d["roofline"]
    print(f.split('/')[-1], d["value"], d["us_per_update"], "critic train", r["avg_launch_us"])
PY
