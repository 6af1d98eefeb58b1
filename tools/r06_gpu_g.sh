#!/bin/bash
# role-split persistent rollout: the rollout tests, then timing with ERL_RF_ROLE_SPLIT = 0 / 1 alternating on one box (c4, c2, cd), phase profiles
O=$GRAFT_REPO_ROOT/gpurun_out/r06_g; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 900 python -m pytest tests/test_rollout_fused_gpu.py tests/test_agent_gpu.py -m gpu -q -x > $O/pytest_rollout.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rollout.log
tail -6 $O/pytest_rollout.log
for cfg in c4 c2 cd; do
  for rep in 0 1; do
    for r in 0 1; do
      ERL_RF_ROLE_SPLIT=$r timeout 300 python bench.py --config $cfg --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/${cfg}_rs${r}_$rep.json 2> $O/${cfg}_rs${r}_$rep.err
    done
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c*_rs?_?.json")):
    try:
        d = json.loads(open(f).readline()); b = d["breakdown"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "explore_ms", b["explore_env_ms"], "update_ms", b["update_net_ms"])
    except Exception as e:
        print(f, "FAILED", e)
PY
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
for r in 0 1; do
  ERL_RF_ROLE_SPLIT=$r ERL_HIP_PROF_LIB=$L/liberl_hip_prof.so python tools/rollout_fused_phase_profile.py > $O/rollout_phase_rs$r.txt 2>&1; tail -11 $O/rollout_phase_rs$r.txt | cut -c1-150
  ERL_RF_ROLE_SPLIT=$r RF_ENV=pendulum ERL_HIP_PROF_LIB=$L/liberl_hip_prof.so python tools/rollout_fused_phase_profile.py > $O/rollout_phase_pend_rs$r.txt 2>&1; tail -11 $O/rollout_phase_pend_rs$r.txt | cut -c1-150
done
