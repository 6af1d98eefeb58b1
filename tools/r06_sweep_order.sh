#!/bin/bash
# what sits between the warm-up and the timed region: round 6 (gc + null bracket + GAE sweep BEFORE the warm-up, collector off) against rounds 3-5
# (gc.collect + 200 empty launches after the warm-up, sweep after the region); alternating processes on ONE box
O=$GRAFT_REPO_ROOT/gpurun_out/r06_order; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
for rep in 0 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-smi > $O/before_$rep.json 2> /dev/null
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-smi --gc-after-warmup --gae-sweep-after > $O/after_$rep.json 2> /dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(f).readline()); b = d["breakdown"]
    print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["shader_mhz"], [round(x, 2) for x in b["update_net_ms_each"][::4]])
PY
