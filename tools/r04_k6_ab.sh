#!/bin/bash
# round 4, K6 A/B on one box: the round-3 library (liberl_hip_r03.so, built from the round-3 sources) against the current one --
# bench.py's config-4 loop under the kernel trace (the judged command), the in-kernel phase profile, the two-wave probe and the
# K6 parity tests.  gpurun -- bash tools/r04_k6_ab.sh [tag]
TAG=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/r04_k6_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
tools/bin/twowave_probe > $O/twowave_probe.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py -m gpu -x -q -k "ppo or update or golden or split" > $O/pytest_k6.log 2>&1; echo "rc=$?" >> $O/pytest_k6.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in r03 new r03 new; do
  lib=$L/liberl_hip.so; [ $v = r03 ] && lib=$L/liberl_hip_r03.so
  [ -f $lib ] || continue
  n=$(ls $O | grep -c "bench_$v")
  ERL_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${v}_$n -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --repeats 2 > $O/bench_${v}_$n.json 2> /dev/null
  python tools/kstats_short.py $O/prof_${v}_$n > $O/kstats_${v}_$n.txt 2>&1
  rm -rf $O/prof_${v}_$n
done
K6_LOOP=1 K6_SHAPE=64,128,128,8 python tools/ppo_phase_profile.py > $O/k6_phase_c4.txt 2>&1
tail -3 $O/pytest_k6.log; grep -h "ppo_step_s3\|reduce_exch\|clip_adam" $O/kstats_*.txt; for f in $O/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['extra']['repeated_regions_ms_per_step'], d['roofline']['avg_launch_us'])"; done
grep "^F\|^G" $O/twowave_probe.txt
