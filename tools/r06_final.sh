#!/bin/bash
# round 6, closing numbers: ONE call on one box -- the whole GPU suite, the box record, every bench configuration, rocprofv3 kernel statistics
# of c4 / cd / c3, the counter passes of the minibatch kernel, the GAE scan, config 3's step and the replay gather.
#   gpurun -- bash tools/r06_final.sh [tag]
TAG=${1:-final}
O=$GRAFT_REPO_ROOT/gpurun_out/r06_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
python tools/box_record.py > $O/box.json 2> $O/box.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
for c in c2 c3 c5 cw cd c1; do python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
python tools/gae_lb_sweep.py 32x4096 128x4096 200x4096 1024x4096 2048x4096 32x32768 > $O/gae_lb_sweep.txt 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 0 > $O/bench_c4_under_rocprof.json 2> /dev/null
cp $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1) $O/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cd -o cd -- python bench.py --config cd --steps 3 --warmup 2 --no-cpu-baseline --no-smi --repeats 0 > $O/bench_cd_under_rocprof.json 2> /dev/null
cp $(find $O/prof_cd -name "*kernel_stats.csv" | head -1) $O/cd_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1) $O/c3_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
python tools/pmc_summarise.py $O/r06_pmc_traffic.json $(find $O/pmc_fetch $O/pmc_write $O/pmc_sq -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  d=$O/pmc_c3_$(echo $c | cut -d' ' -f1)
  C3_PART=step rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python tools/c3_pmc_workload.py > /dev/null 2>&1
done
PMC_KEEP_TEMPLATE=1 python tools/pmc_summarise.py $O/r06_c3_pmc_by_pass.json $(find $O/pmc_c3_* -name "*counter_collection.csv") > $O/pmc_c3_by_pass.txt 2>&1
python tools/pmc_summarise.py $O/r06_c3_pmc.json $(find $O/pmc_c3_* -name "*counter_collection.csv") > $O/pmc_c3.txt 2>&1
python tools/kstats_summarise.py $O/r06_kernel_times.json $O/c4_kernel_stats.csv "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 0" > $O/kernel_times.txt 2>&1
rm -rf $O/prof_c4 $O/prof_cd $O/prof_c3 $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_c3_*
ERL_HIP_PROF_LIB=$GRAFT_REPO_ROOT/elegantrl_amd/lib/liberl_hip_prof.so K6_LOOP=1 python tools/ppo_phase_profile.py > $O/k6_phase_c4.txt 2>&1
ERL_HIP_PROF_LIB=$GRAFT_REPO_ROOT/elegantrl_amd/lib/liberl_hip_prof.so K6_LOOP=1 K6_SHAPE=3,128,64,1 python tools/ppo_phase_profile.py > $O/k6_phase_c2.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python - <<PY
import json
O = "$O"
for c in ("c4", "c2", "c3", "c5", "cw", "cd", "c1"):
    try:
        d = json.loads(open(f"{O}/bench_{c}.json").readline())
        r = d["roofline"]
        print(c, d["value"], d["unit"], "ms/step", d["ms_per_step"], "steady", (d.get("extra") or {}).get("repeated_regions_ms_per_step"), "roofline",
              {k: r.get(k) for k in ("kernel", "avg_launch_us", "frac", "traffic", "box_ratio", "shader_mhz", "instruction_fetch")}, "gae", (d.get("roofline_gae") or {}).get("frac"),
              (d.get("roofline_gae") or {}).get("avg_launch_us"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "breakdown", {k: v for k, v in (d.get("breakdown") or {}).items() if k.endswith("_us") or k.endswith("_ms")})
    except Exception as e:
        print(c, "FAILED", e, open(f"{O}/bench_{c}.err").read()[-400:])
PY
grep -E "ppo_step_s3|reduce_exch|clip_adam_partials|rollout_fused|gae_" $O/c4_kernel_stats.csv $O/cd_kernel_stats.csv | cut -c1-230
grep -E "critic_tile|actor_fwd|actor_bwd|dw_table|replay_sample" $O/c3_kernel_stats.csv | cut -c1-200
tail -5 $O/kernel_times.txt
