#!/bin/bash
# layered path with 128-wide GEMM tiles: tests + A/B of the tile shapes
O=$GRAFT_REPO_ROOT/gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_mlpn_gpu.py tests/test_discrete_gpu.py tests/test_sac.py -m gpu -q -x 2>&1 | tail -3
for t in 0 11 12 21 22; do
  for h in 128,128 256,128 256,128,64; do echo -n "tile=$t "; ERL_GEMM_TILE=$t python tools/mlpn_step_profile.py $h 2>&1 | tail -1; done
done | tee $O/mlpn_tiles.txt
