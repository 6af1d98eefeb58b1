#!/bin/bash
# the round's closing numbers after the last kernel changes (rollout on the bf16 pipe, SAC tail / split / persistent rollout, wide rollout step):
# gpurun -- bash tools/r04_final.sh; results under gpurun_out/r04_last/, copied into profiles/ by hand afterwards.
O=$GRAFT_REPO_ROOT/gpurun_out/r04_last; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_c4.json 2> $O/bench_c4.err
for c in c2 c3 c5 cw; do python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
python tools/rollout_fused_phase_profile.py > $O/rollout_fused_phase_c4.txt 2>&1
RF_ENV=pendulum python tools/rollout_fused_phase_profile.py > $O/rollout_fused_phase_c2.txt 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --repeats 0 > $O/bench_c4_under_rocprof.json 2> /dev/null
cp $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1) $O/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1) $O/c3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof_c2 -name "*kernel_stats.csv" | head -1) $O/c2_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cw -o cw -- python bench.py --config cw --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof_cw -name "*kernel_stats.csv" | head -1) $O/cw_kernel_stats.csv
python tools/kstats_summarise.py $O/r04_kernel_times.json $O/c4_kernel_stats.csv "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --repeats 0" > $O/kernel_times.txt 2>&1
rm -rf $O/prof_c4 $O/prof_c3 $O/prof_c2 $O/prof_cw
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python - <<'PY'
import json,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04_last"
for c in ("c4","c2","c3","c5","cw"):
    try:
        d=json.loads(open(f"{O}/bench_{c}.json").readline())
        print(c, d["value"], d["unit"], "ms/step", d["ms_per_step"], "steady", (d.get("extra") or {}).get("repeated_regions_ms_per_step"), "roofline", {k:d["roofline"].get(k) for k in ("kernel","avg_launch_us","frac","traffic")}, "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(c, "FAILED", e, open(f"{O}/bench_{c}.err").read()[-400:])
PY
