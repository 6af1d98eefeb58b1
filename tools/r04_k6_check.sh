#!/bin/bash
# K6 edit loop on the GPU box: parity tests of everything that runs through the minibatch kernels, the in-kernel phase profile,
# one bench region.  gpurun -- bash tools/r04_k6_check.sh [full]
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py -m gpu -x -q -k "ppo or update or golden or split" 2>&1 | tail -5
K6_LOOP=1 python tools/ppo_phase_profile.py 2>&1 | grep -v amdgpu.ids | head -24
python bench.py --no-cpu-baseline --no-gae-sweep --repeats 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('bench', d['value'], d['ms_per_step'], d['extra']['repeated_regions_ms_per_step'], 'k6 span', d['roofline']['avg_launch_us'], 'rest/minibatch', d['breakdown']['per_minibatch_rest_us'], 'explore', d['breakdown']['explore_env_ms'])"
