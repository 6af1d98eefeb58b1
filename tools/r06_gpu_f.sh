#!/bin/bash
# what the critic's layers cost the persistent rollout's per-step chain: the step's phase profile with and without them (timing only)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_f; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
for v in prof rfnc; do
  ERL_HIP_PROF_LIB=$L/liberl_hip_$v.so python tools/rollout_fused_phase_profile.py > $O/rollout_phase_$v.txt 2>&1; tail -11 $O/rollout_phase_$v.txt | cut -c1-150
  RF_ENV=pendulum ERL_HIP_PROF_LIB=$L/liberl_hip_$v.so python tools/rollout_fused_phase_profile.py > $O/rollout_phase_pend_$v.txt 2>&1; tail -11 $O/rollout_phase_pend_$v.txt | cut -c1-150
done
