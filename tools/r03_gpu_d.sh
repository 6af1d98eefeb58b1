#!/bin/bash
# round 3, GPU call D: full GPU suite + K6 phase profiles (config 2 / config 4 shapes)
O=gpurun_out/r03d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
K6_SHAPE=3,128,64,1 timeout 300 python tools/ppo_phase_profile.py > $O/k6_phase_c2.txt 2>&1
K6_SHAPE=64,128,128,8 timeout 300 python tools/ppo_phase_profile.py > $O/k6_phase_c4.txt 2>&1
tail -8 $O/pytest.log; cat $O/k6_phase_c2.txt
