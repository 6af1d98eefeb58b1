#!/bin/bash
# round 4, call b: new timing hooks in bench.py, world-4/8 exchange tests, prologue experiments (tools/build_variants.sh variants)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_b; mkdir -p $O
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
timeout 1500 python -m pytest tests/test_parallel_gpu.py -m gpu -x -q > $O/pytest_parallel.log 2>&1; echo "rc=$?" >> $O/pytest_parallel.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py tests/test_bench_gpu.py -m gpu -x -q > $O/pytest_k.log 2>&1; echo "rc=$?" >> $O/pytest_k.log
python bench.py --no-cpu-baseline --no-gae-sweep --repeats 2 > $O/bench_new.json 2> $O/bench_new.err
ERL_HIP_LIB=$L/liberl_hip_x2.so python bench.py --no-cpu-baseline --no-gae-sweep --repeats 2 > $O/bench_x2.json 2> $O/bench_x2.err
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_new -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --repeats 0 > $O/bench_new_rocprof.json 2> /dev/null
python tools/kstats_short.py $O/prof_new > $O/kstats_new.txt 2>&1; rm -rf $O/prof_new
for v in prof prof_e1 prof_e2; do
  ERL_HIP_PROF_LIB=$L/liberl_hip_$v.so K6_LOOP=1 python tools/ppo_phase_profile.py > $O/phase_$v.txt 2>&1
done
ERL_HIP_PROF_LIB=$L/liberl_hip_prof.so K6_LOOP=1 K6_IDS=contiguous python tools/ppo_phase_profile.py > $O/phase_prof_contig.txt 2>&1
tail -4 $O/pytest_parallel.log; tail -3 $O/pytest_k.log
for f in new x2 new_rocprof; do python - <<PY
import json
try:
    d=json.load(open("$O/bench_$f.json")); r=d["roofline"]
    print("$f", d["value"], d["ms_per_step"], d.get("extra",{}).get("repeated_regions_ms_per_step"), "span", r["avg_launch_us"], "event", r["event_bracket_us"], "null", r["event_bracket_null_us"], "frac", r["frac"], d["breakdown"])
except Exception as e: print("$f", "failed", e)
PY
done
grep -h "ppo_step_s3\|reduce_exch\|clip_adam\|rollout" $O/kstats_new.txt
for v in prof prof_e1 prof_e2 prof_contig; do echo "== $v"; grep -A4 "^--- actor" $O/phase_$v.txt | cut -c1-110; grep "wall time" $O/phase_$v.txt; done
