#!/bin/bash
# two-chain update loop: bitwise check + loop time, then config 4 with ERL_PPO_CHAINS = 1 / 2 alternating on one box
O=$GRAFT_REPO_ROOT/gpurun_out/r06_chains; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 300 python tools/chains_check.py 40 > $O/chains_check.txt 2>&1; cat $O/chains_check.txt | tail -14
for rep in 0 1; do
  for c in 1 2; do
    ERL_PPO_CHAINS=$c timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/c${c}_$rep.json 2> $O/c${c}_$rep.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c?_?.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "update_ms", b["update_net_ms"], "reduce", b["slab_reduce_us"], "adam", b["clip_adam_us"])
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 $O/c2_0.err
