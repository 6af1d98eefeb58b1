#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r06_n; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py tests/test_bench_gpu.py -m gpu -q -x -k "gae or adv or agent or golden or bench" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
d = json.loads(open("$O/bench_c2.json").readline())
print(d["value"], d["ms_per_step"], d.get("roofline_gae"))
PY
