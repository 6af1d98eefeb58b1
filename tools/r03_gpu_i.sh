#!/bin/bash
O=gpurun_out/r03i; mkdir -p $O
python tools/sac_fused_profile.py > $O/sac_fused_profile.txt 2>&1
timeout 900 python -m pytest tests/test_sac.py -m gpu -q -x > $O/pytest_sac.log 2>&1
echo "pytest rc=$?" >> $O/pytest_sac.log
timeout 300 python bench.py --config c3 > $O/bench_c3_fused.json 2> $O/bench_c3_fused.err
cat $O/sac_fused_profile.txt | tail -10; tail -3 $O/pytest_sac.log; cut -c1-200 $O/bench_c3_fused.json
