#!/bin/bash
# round 5, fourth call: box record with per-workgroup placement / cold-cache / in-loop K6, the 8-rank rehearsal tests, bench c4.
TAG=${1:-d}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
python tools/box_record.py > $O/box.json 2> $O/box.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "rehearsal or falls_back or two_ranks or single_launch or lookback" > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log
tail -5 $O/pytest_sel.log
python - <<PY
import json
d = json.loads(open("$O/bench_c4.json").readline())
r = d["roofline"]
print("c4", d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], r["avg_launch_us"], r["frac"], r["box_ratio"], r["shader_mhz"], r["phase_cycles"])
print("  wg:", d["clocks"]["k6_workgroups_last_sampled_launch"])
print("  breakdown:", {k: v for k, v in d["breakdown"].items() if not k.endswith("each") and k != "note"})
b = json.load(open("$O/box.json"))
for k, v in b["k6_standalone"].items():
    print("box", k, {kk: vv for kk, vv in v.items() if kk not in ("note",)})
print(b.get("clock_probe"))
PY
