#!/bin/bash
# config 3: the target pass folded into the split training pass (default) against a launch of its own; alternating processes, ONE box
O=$GRAFT_REPO_ROOT/gpurun_out/r06_sacx; mkdir -p $O; rm -f $O/*.json
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 900 python -m pytest tests/test_sac.py -m gpu -q -x > $O/pytest_sac.log 2>&1; tail -2 $O/pytest_sac.log
for rep in 0 1; do
  python bench.py --config c3 --no-cpu-baseline > $O/c3_fold_$rep.json 2> /dev/null
  ERL_SAC_FOLD_TARGET=0 python bench.py --config c3 --no-cpu-baseline > $O/c3_nofold_$rep.json 2> /dev/null
  ERL_SAC_FORK=2 python bench.py --config c3 --no-cpu-baseline > $O/c3_round5_$rep.json 2> /dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c3_*.json")):
    d = json.loads(open(f).readline()); r = d["roofline"]
    print(f.split('/')[-1], d["value"], d["us_per_update"], "critic train", r["avg_launch_us"])
PY
