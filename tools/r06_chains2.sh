#!/bin/bash
# two-chain update loop as the default: the whole GPU suite, then configs c4 / c2 / c5 / cd with ERL_PPO_CHAINS = 1 / 2 alternating on one box
O=$GRAFT_REPO_ROOT/gpurun_out/r06_chains2; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for cfg in c4 c2 c5 cd; do
  for rep in 0 1; do
    for c in 1 2; do
      ERL_PPO_CHAINS=$c timeout 300 python bench.py --config $cfg --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/${cfg}_c${c}_$rep.json 2> $O/${cfg}_c${c}_$rep.err
    done
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c*_c?_?.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "chains", b.get("update_loop_chains"), "k6", r["avg_launch_us"], "frac", r["frac"], "update_ms", b["update_net_ms"], "reduce", b["slab_reduce_us"], "adam", b["clip_adam_us"], (r.get("by_position_in_the_update_loop") or {}).get("actor_launches_us"), (r.get("by_position_in_the_update_loop") or {}).get("critic_launches_us"))
    except Exception as e:
        print(f, "FAILED", e)
PY
