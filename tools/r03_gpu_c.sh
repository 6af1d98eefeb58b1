#!/bin/bash
# round 3, GPU call C: the generalised one-wave-per-SIMD K6 ([128,64] etc., unaligned S): full GPU suite, c4 (no regression), c2 A/B
O=gpurun_out/r03c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_c4.json 2> $O/bench_c4.err
timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
ERL_K6_FORM=8 timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2_form8.json 2> $O/bench_c2_form8.err
timeout 300 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
tail -5 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c/bench_*.json")):
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d.get("extra",{}).get("repeated_regions_ms_per_step"))
PY
