#!/bin/bash
# round 3, GPU call A: the new optimiser tail + exchange routes (tests, tail micro-bench, bench with/without the DP branch)
O=gpurun_out/r03a; mkdir -p $O
timeout 1500 python -m pytest tests/test_parallel_gpu.py tests/test_kernels_gpu.py -k "tail or p2p or one_rank or lockstep or clip_adam or update_loop" -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/tail_bench.py > $O/tail_bench.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_plain.json 2> $O/bench_plain.err
for m in auto rccl p2p torch; do
  ERL_FORCE_DP=1 ERL_DP_COLLECTIVE=$m timeout 300 python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_dp_$m.json 2> $O/bench_dp_$m.err
done
timeout 300 python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_plain2.json 2> $O/bench_plain2.err
tail -3 $O/pytest.log; cat $O/tail_bench.txt; cat $O/bench_*.json | cut -c1-400
