#!/bin/bash
# SAC: launch (1) and the policy-gradient sample's forward as one launch (ERL_SAC_PAIR=1, actor_fwd_pair_kernel) against the side stream (0): SAC tests both ways, c3 alternating
O=$GRAFT_REPO_ROOT/gpurun_out/r06_pair; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
for p in 1 0; do
  ERL_SAC_PAIR=$p timeout 900 python -m pytest tests/test_sac.py tests/test_per.py -m gpu -q -x > $O/pytest_pair$p.log 2>&1; echo "pytest pair=$p rc=$?" >> $O/pytest_pair$p.log
  tail -2 $O/pytest_pair$p.log
done
for rep in 0 1 2; do
  for p in 0 1; do
    ERL_SAC_PAIR=$p timeout 300 python bench.py --config c3 --no-cpu-baseline > $O/c3_pair${p}_$rep.json 2> $O/c3_pair${p}_$rep.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c3_pair*_?.json")):
    try:
        d = json.loads(open(f).readline()); print(f.split('/')[-1], d["value"], d["us_per_update"], d["roofline"]["avg_launch_us"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-300:])
PY
