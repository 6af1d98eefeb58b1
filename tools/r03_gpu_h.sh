#!/bin/bash
# round 3, GPU call H: fused SAC step after the branch-free restructure: tests, bench, per-kernel times
O=gpurun_out/r03h; mkdir -p $O
timeout 900 python -m pytest tests/test_sac.py -m gpu -q -x > $O/pytest_sac.log 2>&1
echo "pytest rc=$?" >> $O/pytest_sac.log
timeout 300 python bench.py --config c3 > $O/bench_c3_fused.json 2> $O/bench_c3_fused.err
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err
cp $O/prof/c3_kernel_stats.csv $O/c3_fused_kernel_stats.csv 2>/dev/null
tail -4 $O/pytest_sac.log
cut -c1-220 $O/bench_c3_fused.json
head -16 $O/c3_fused_kernel_stats.csv | cut -c1-150
