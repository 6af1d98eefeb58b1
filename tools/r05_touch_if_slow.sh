#!/bin/bash
# one box: the clock probe's code walk says whether its instruction fetch is slow (two 56 KB blocks per cache > 35 us); only then the
# code-touch A/B of tools/r05_touch.sh runs (slow boxes are ~1 in 4 of the pool; a fast box costs this script ~8 s)
TAG=${1:-s}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_touch_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
tools/bin/clock_probe > $O/clock_probe.json 2> $O/clock_probe.err
SLOW=$(python -c "import json; print(1 if json.load(open('$O/clock_probe.json'))['code_walk_56KB']['dual_map0']['back_to_back_us'] > 35 else 0)")
echo "slow=$SLOW"
[ "$SLOW" = 1 ] && bash tools/r05_touch.sh $TAG skip
