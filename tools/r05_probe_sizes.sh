#!/bin/bash
# one box: the clock probe; prints the code walks and the capacity sweep.  gpurun -- bash tools/r05_probe_sizes.sh <tag>
TAG=${1:-z}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_probe_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
tools/bin/clock_probe > $O/clock_probe.json 2> $O/clock_probe.err
python - <<PY
import json
cp = json.load(open("$O/clock_probe.json"))
w = cp["code_walk_56KB"]
print("slow=%d" % (1 if w["dual_map0"]["back_to_back_us"] > 35 else 0), {k: v["back_to_back_us"] for k, v in w.items()})
s = cp["code_walk_us_by_KB"]; print("us_by_KB", s); print("us_per_KB", {k: round(v / int(k), 3) for k, v in s.items()})
PY
