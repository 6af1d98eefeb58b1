#!/bin/bash
# whole GPU suite on the working tree (two-chain loop built, default off) + the persistent rollout's phase profile
O=$GRAFT_REPO_ROOT/gpurun_out/r06_e; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python tools/rollout_fused_phase_profile.py > $O/rollout_phase.txt 2>&1; cat $O/rollout_phase.txt | cut -c1-200
