#!/bin/bash
# the round's last check on the committed tree: smoke(), the whole GPU suite, the driver's bench command
O=$GRAFT_REPO_ROOT/gpurun_out/r06_verify; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
python -c "
import json
d=json.loads(open('$O/bench_c4.json').readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], d['extra']['repeated_regions_ms_per_step'], r['avg_launch_us'], r['frac'], r['traffic'], r.get('box_ratio'), r.get('instruction_fetch'), d['roofline_gae'].get('sweep_ran'))
"
