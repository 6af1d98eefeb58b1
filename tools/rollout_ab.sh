#!/bin/bash
# A/B of two library builds on ONE box: ERL_HIP_LIB=elegantrl_amd/lib/liberl_hip_old.so (the previous commit, built from `git archive`) against
# the current one -- the rollout's arithmetic difference (tools/rollout_arith_ab.py) and bench.py configs 4 and 2, alternating.
O=gpurun_out/rollout_ab; mkdir -p $O
OLD=$PWD/elegantrl_amd/lib/liberl_hip_old.so
ERL_HIP_LIB=$OLD python tools/rollout_arith_ab.py old 2>/dev/null | tee $O/arith.txt
python tools/rollout_arith_ab.py new 2>/dev/null | tee -a $O/arith.txt
python tools/rollout_arith_ab.py --compare old new | tee -a $O/arith.txt
rm -f gpurun_out/rollout_ab_old.npz gpurun_out/rollout_ab_new.npz
for round in 1 2; do
  for c in c4 c2; do
    ERL_HIP_LIB=$OLD python bench.py --config $c --repeats 2 2>/dev/null > $O/${c}_old_$round.json
    python bench.py --config $c --repeats 2 2>/dev/null > $O/${c}_new_$round.json
  done
done
python - <<'PY' | tee $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/rollout_ab/c?_*.json")):
    d=json.loads(open(f).readline())
    print(f.split("/")[-1], "value %.4g" % d["value"], "ms %.3f" % d["ms_per_step"], "explore %.4f" % d["breakdown"]["explore_env_ms"],
          "steady", d["extra"]["repeated_regions_ms_per_step"], "objs", d["objectives_last"])
PY
