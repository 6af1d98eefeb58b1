#!/bin/bash
# round 6, last session: dw_table_kernel forms (ERL_SAC_DW) -- the bitwise test, then config 3 per form, three times each, and kernel statistics of forms 1 / 0
#   gpurun -- bash tools/r06_sac_dw.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r06_sac_dw; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
python -m pytest tests/test_sac.py -m gpu -x -q 2>&1 | tail -3
for rep in 0 1 2; do for f in 1 0 5 6; do
  ERL_SAC_DW=$f python bench.py --config c3 --no-cpu-baseline > $O/c3_dw${f}_$rep.json 2> $O/c3_dw${f}_$rep.err
done; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 0 5; do
  ERL_SAC_DW=$f rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$f -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  cp $(find $O/prof_$f -name "*kernel_stats.csv" | head -1) $O/c3_dw${f}_kernel_stats.csv; rm -rf $O/prof_$f
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c3_dw*_?.json")):
    try:
        d = json.loads(open(f).readline()); print(os.path.basename(f), d["value"], d["us_per_update"])
    except Exception as e:
        print(f, "FAILED", e)
PY
grep -h "dw_table" $O/c3_dw0_kernel_stats.csv $O/c3_dw5_kernel_stats.csv | cut -c1-160
