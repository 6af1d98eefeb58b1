#!/bin/bash
# config 2 (200 x 4096): get_advantages in the rollout's epilogue (ERL_FUSED_GAE=1, default) against the scan kernels (0: since round 6 one workgroup per 32 envs)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_fg; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
for rep in 0 1 2; do
  for f in 1 0; do
    ERL_FUSED_GAE=$f timeout 300 python bench.py --config c2 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/c2_fg${f}_$rep.json 2> $O/c2_fg${f}_$rep.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c2_fg*_?.json")):
    try:
        d = json.loads(open(f).readline()); b = d["breakdown"]; g = d.get("roofline_gae") or {}
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "explore", b["explore_env_ms"], "update", b["update_net_ms"], "gae", g.get("avg_launch_us"), g.get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
