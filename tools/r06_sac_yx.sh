#!/bin/bash
# round 6, last session: the head shares' meeting in the split actor forward -- owner form (ERL_SAC_YX=1, granules + the last-dispatched slice polls) against
# the last-arriver form (=0); tests under both, config 3 alternating, kernel statistics
#   gpurun -- bash tools/r06_sac_yx.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r06_sac_yx; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
for y in 1 0; do ERL_SAC_YX=$y python -m pytest tests/test_sac.py -m gpu -x -q 2>&1 | tail -1; done
for rep in 0 1 2; do for y in 0 1; do
  ERL_SAC_YX=$y python bench.py --config c3 --no-cpu-baseline > $O/c3_yx${y}_$rep.json 2> $O/c3_yx${y}_$rep.err
done; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for y in 0 1; do
  ERL_SAC_YX=$y rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$y -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  cp $(find $O/prof_$y -name "*kernel_stats.csv" | head -1) $O/c3_yx${y}_kernel_stats.csv; rm -rf $O/prof_$y
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c3_yx?_?.json")):
    d = json.loads(open(f).readline()); print(os.path.basename(f), d["value"], d["us_per_update"], d["objectives_last"])
PY
grep -h "actor_fwd_pair" $O/c3_yx0_kernel_stats.csv $O/c3_yx1_kernel_stats.csv | cut -c1-200
python -c "
import sys; sys.path.insert(0, '.')
from elegantrl_amd import _hip; import torch
print('async faults', _hip.lib().erl_async_fault_count(0))"
