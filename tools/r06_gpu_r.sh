#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r06_r; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 1200 python -m pytest tests/test_rollout_fused_gpu.py tests/test_agent_gpu.py tests/test_bench_gpu.py tests/test_compat.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
d = json.loads(open("$O/bench_c2.json").readline()); b = d["breakdown"]; g = d["roofline_gae"]
print(d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], b["explore_env_ms"], b["update_net_ms"], {k: g.get(k) for k in ("kernel", "avg_launch_us", "frac", "achieved")}, d["cpu_baseline"]["value"])
PY
