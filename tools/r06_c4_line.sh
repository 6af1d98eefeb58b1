#!/bin/bash
# the driver's command once more, AFTER the counter / rocprofv3 summaries of tools/r06_final.sh were copied into profiles/ (so that the line carries
# `traffic` and `box_ratio` from files stamped with the sources' hash)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_line; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err
python -c "
import json
d=json.loads(open('$O/bench_c4.json').readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], d['extra']['repeated_regions_ms_per_step'], r['avg_launch_us'], r['frac'], r['traffic'], r.get('box_ratio'), r.get('instruction_fetch'))
print(json.dumps(d['roofline_gae']['sweep'])[:1500])
print(json.dumps(d['cpu_baseline'])[:400])
"
