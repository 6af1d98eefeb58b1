#!/bin/bash
# GAE: one workgroup per 32 envs for 64 <= H <= 512 (gae_tall_kernel): the GAE / agent tests, then the sweep with it off and on
O=$GRAFT_REPO_ROOT/gpurun_out/r06_m; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py -m gpu -q -x -k "gae or adv or agent or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for t in 0 1; do
  ERL_GAE_TALL=$t python tools/gae_lb_sweep.py 64x4096 128x4096 200x4096 256x4096 512x4096 128x32768 256x16384 200x512 2>&1 | grep '"L": null' | grep lookback > $O/sweep_tall$t.txt
  echo "== ERL_GAE_TALL=$t"; cat $O/sweep_tall$t.txt | cut -c1-250
done
