#!/bin/bash
# round 6, last session: SAC tile kernels -- shares / partial values / table entries that were fetched one round trip at a time are requested together
#   gpurun -- bash tools/r06_sac_batch.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r06_sac_batch; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
python -m pytest tests/test_sac.py -m gpu -x -q 2>&1 | tail -2
for rep in 0 1 2; do python bench.py --config c3 --no-cpu-baseline > $O/c3_$rep.json 2> $O/c3_$rep.err; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/c3_kernel_stats.csv; rm -rf $O/prof
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c3_?.json")):
    d = json.loads(open(f).readline()); print(os.path.basename(f), d["value"], d["us_per_update"], d["objectives_last"])
PY
grep -E "critic_tile|actor_fwd|actor_bwd|dw_table|clip_adam" $O/c3_kernel_stats.csv | awk -F'","|",' '{print $1, $2, $4}' | cut -c1-150
