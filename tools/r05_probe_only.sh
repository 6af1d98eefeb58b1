#!/bin/bash
# one box, the clock probe only (~6 s): where the kernel code lives, the 56 KB code walks, the single-pipe bodies.  gpurun -- bash tools/r05_probe_only.sh <tag>
TAG=${1:-q}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_probe_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
tools/bin/clock_probe > $O/clock_probe.json 2> $O/clock_probe.err
python - <<PY
import json
cp = json.load(open("$O/clock_probe.json"))
print("code_memory", cp.get("code_memory"))
for k, v in cp.get("code_walk_56KB", {}).items(): print("code_walk", k, v)
PY
