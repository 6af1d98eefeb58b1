#!/bin/bash
# round 3, GPU call G: per-kernel times of the fused SAC step (rocprofv3 kernel trace of bench.py --config c3)
O=gpurun_out/r03g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/c3_fused_kernel_stats.csv 2>/dev/null
head -30 $O/c3_fused_kernel_stats.csv | cut -c1-200
