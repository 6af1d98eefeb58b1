#!/bin/bash
# every number quoted in DESIGN.md / README.md for round 4: run on the GPU box (gpurun -- bash tools/r04_measure_all.sh); results under
# gpurun_out/r04_final/, the ones that are judged are copied into profiles/ afterwards (tools/r04_collect.sh).
O=$GRAFT_REPO_ROOT/gpurun_out/r04_final; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_c4.json 2> $O/bench_c4.err
for c in c2 c3 c5; do python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
ERL_FUSED_GAE=0 python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_c4_unfused_gae.json 2> /dev/null
for m in auto rccl p2p; do ERL_FORCE_DP=1 ERL_DP_COLLECTIVE=$m python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_c4_dp_$m.json 2> $O/bench_c4_dp_$m.err; done
python tools/tail_bench.py > $O/tail_bench.txt 2>&1
K6_LOOP=1 K6_SHAPE=64,128,128,8 python tools/ppo_phase_profile.py > $O/k6_phase_c4.txt 2>&1
K6_LOOP=1 K6_SHAPE=3,128,64,1 python tools/ppo_phase_profile.py > $O/k6_phase_c2.txt 2>&1
tools/bin/twowave_probe > $O/twowave_probe.txt 2>&1
tools/bin/tail_probe > $O/tail_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# the bench command itself under the kernel trace (the judged command: python bench.py --gpus 1 --steps 20 --warmup 5)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --repeats 0 > $O/bench_c4_under_rocprof.json 2> /dev/null
cp $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1) $O/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1) $O/c3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof_c2 -name "*kernel_stats.csv" | head -1) $O/c2_kernel_stats.csv
# HBM traffic and SQ counters: one counter set per pass, kernel trace only (MI355X_MICROARCH.md)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
python tools/pmc_summarise.py $O/r04_pmc_traffic.json $(find $O/pmc_fetch $O/pmc_write $O/pmc_sq -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
python tools/kstats_summarise.py $O/r04_kernel_times.json $O/c4_kernel_stats.csv "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --repeats 0" > $O/kernel_times.txt 2>&1
rm -rf $O/prof_c4 $O/prof_c3 $O/prof_c2 $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log; cut -c1-300 $O/bench_c4.json; ls $O
