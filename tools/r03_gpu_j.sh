#!/bin/bash
# round 3, GPU call J: K6 tiny-S path (S <= 8): gradient tests, config 2 bench, config 4 sanity
O=gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py -m gpu -q -x -k "ppo_step or update_loop or agent or iteration or pendulum" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_c4.json 2> $O/bench_c4.err
tail -4 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03j/bench_*.json")):
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d.get("extra",{}).get("repeated_regions_ms_per_step"))
PY
