#!/bin/bash
# SAC paired launch: side stream (0) / pair + temperature step on the side stream (1) / pair, no side stream at all (2); then rocprofv3 kernel statistics of 1 and 2
O=$GRAFT_REPO_ROOT/gpurun_out/r06_pair2; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
for p in 4 3; do ERL_SAC_PAIR=$p timeout 900 python -m pytest tests/test_sac.py tests/test_per.py -m gpu -q -x > $O/pytest_pair$p.log 2>&1; echo "pytest pair=$p rc=$?" >> $O/pytest_pair$p.log; tail -2 $O/pytest_pair$p.log; done
for rep in 0 1 2; do
  for p in 0 3 4; do
    ERL_SAC_PAIR=$p timeout 300 python bench.py --config c3 --no-cpu-baseline > $O/c3_pair${p}_$rep.json 2> $O/c3_pair${p}_$rep.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c3_pair*_?.json")):
    try:
        d = json.loads(open(f).readline()); print(f.split('/')[-1], d["value"], d["us_per_update"], d["roofline"]["avg_launch_us"])
    except Exception as e:
        print(f, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for p in 4; do
  ERL_SAC_PAIR=$p rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$p -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  cp $(find $O/prof$p -name "*kernel_stats.csv" | head -1) $O/c3_pair${p}_kernel_stats.csv; rm -rf $O/prof$p
  echo "== pair=$p"; head -12 $O/c3_pair${p}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-260
done
