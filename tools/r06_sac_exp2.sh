#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r06_sacx2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
for v in early round5; do
  if [ $v = round5 ]; then export ERL_SAC_FORK=2; else unset ERL_SAC_FORK; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "== $v"; f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e3, 2), r["Percentage"])
PY
done
