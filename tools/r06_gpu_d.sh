#!/bin/bash
# round 6, fourth call: the split critic training pass of the fused SAC step -- its tests, config 3 with and without it (alternating).
O=$GRAFT_REPO_ROOT/gpurun_out/r06_d; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 900 python -m pytest tests/test_sac.py tests/test_per.py -m gpu -q -x > $O/pytest_sac.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sac.log
tail -5 $O/pytest_sac.log
for rep in 0 1; do
  python bench.py --config c3 --no-cpu-baseline > $O/bench_c3_split_$rep.json 2> $O/bench_c3_split_$rep.err
  ERL_SAC_TRAIN_SPLIT=0 python bench.py --config c3 --no-cpu-baseline > $O/bench_c3_unsplit_$rep.json 2> $O/bench_c3_unsplit_$rep.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_c3_*.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]
        print(f.split('/')[-1], d["value"], d["us_per_update"], r["avg_launch_us"], r["frac"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-600:])
PY
