#!/bin/bash
# a slow-fetch box (tools/clock_probe.hip's code walk says so), every bench configuration on it with the library's defaults
TAG=${1:-s}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_slowcfg_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
tools/bin/clock_probe > $O/clock_probe.json 2> $O/clock_probe.err
SLOW=$(python -c "import json; print(1 if json.load(open('$O/clock_probe.json'))['code_walk_56KB']['dual_map0']['back_to_back_us'] > 35 else 0)")
echo "slow=$SLOW"
[ "$SLOW" = 1 ] || exit 0
python bench.py --gpus 1 --steps 20 --warmup 5 --no-gae-sweep > $O/bench_c4.json 2> $O/bench_c4.err
for c in c5 c2 c3 cw cd; do python bench.py --config $c --no-cpu-baseline --repeats 2 > $O/bench_$c.json 2> $O/bench_$c.err; done
python - <<PY
import json
for c in ("c4", "c5", "c2", "c3", "cw", "cd"):
    d = json.loads(open("$O/bench_%s.json" % c).readline()); r = d["roofline"]
    print(c, d["value"], d["ms_per_step"], (d.get("extra") or {}).get("repeated_regions_ms_per_step"), r.get("avg_launch_us"), r.get("workgroup_map"), r.get("by_position_in_the_update_loop"))
PY
