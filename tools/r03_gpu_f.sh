#!/bin/bash
# round 3, GPU call F: fused SAC step (tests, config-3 bench fused vs layered)
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_sac.py tests/test_per.py -m gpu -q -x > $O/pytest_sac.log 2>&1
echo "pytest rc=$?" >> $O/pytest_sac.log
timeout 300 python bench.py --config c3 > $O/bench_c3_fused.json 2> $O/bench_c3_fused.err
ERL_SAC_FUSED=0 timeout 300 python bench.py --config c3 > $O/bench_c3_layered.json 2> $O/bench_c3_layered.err
tail -25 $O/pytest_sac.log
cat $O/bench_c3_fused.json | cut -c1-300; cat $O/bench_c3_layered.json | cut -c1-300; tail -3 $O/bench_c3_fused.err
