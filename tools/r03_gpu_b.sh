#!/bin/bash
# round 3, GPU call B: exchange kernel with explicit sc0 sc1 accesses (no per-wave system fences)
O=gpurun_out/r03b; mkdir -p $O
timeout 1500 python -m pytest tests/test_parallel_gpu.py tests/test_kernels_gpu.py -k "tail or p2p or one_rank or lockstep or clip_adam or update_loop" -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/tail_bench.py > $O/tail_bench.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_plain.json 2> $O/bench_plain.err
for m in auto rccl p2p; do
  ERL_FORCE_DP=1 ERL_DP_COLLECTIVE=$m timeout 300 python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_dp_$m.json 2> $O/bench_dp_$m.err
done
tail -3 $O/pytest.log; cat $O/tail_bench.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03b/bench_*.json")):
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); print(f, d["ms_per_step"], d["roofline"]["avg_launch_us"], json.dumps(d.get("allreduce")), d.get("extra",{}).get("repeated_regions_ms_per_step"))
PY
