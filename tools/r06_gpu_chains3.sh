#!/bin/bash
# the two-chain update loop (ERL_PPO_CHAINS=2) against one chain on the round's final build: alternating processes on one box
TAG=${1:-ch}; O=$GRAFT_REPO_ROOT/gpurun_out/r06_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
for rep in 0 1 2; do
  for c in 1 2; do
    ERL_PPO_CHAINS=$c timeout 300 python bench.py --config c4 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/c4_ch${c}_$rep.json 2> $O/c4_ch${c}_$rep.err
  done
done
python tools/box_record.py > $O/box.json 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c4_ch*_?.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], r.get("instruction_fetch"), r.get("shader_mhz"), "update_ms", b["update_net_ms"])
    except Exception as e:
        print(f, "FAILED", e)
PY
