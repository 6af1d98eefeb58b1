#!/bin/bash
# every number quoted in DESIGN.md for round 3: run on the GPU box (gpurun -- bash tools/r03_measure_all.sh); results under
# gpurun_out/r03_final/, the ones that are judged are copied into profiles/ afterwards.
O=$GRAFT_REPO_ROOT/gpurun_out/r03_final; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python bench.py > $O/bench_c4.json 2> $O/bench_c4.err
for c in c2 c3 c5; do python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
ERL_SAC_FUSED=0 python bench.py --config c3 > $O/bench_c3_layered.json 2> $O/bench_c3_layered.err
for m in auto rccl p2p; do ERL_FORCE_DP=1 ERL_DP_COLLECTIVE=$m python bench.py --no-cpu-baseline --no-gae-sweep > $O/bench_c4_dp_$m.json 2> $O/bench_c4_dp_$m.err; done
python tools/tail_bench.py > $O/tail_bench.txt 2>&1
K6_LOOP=1 K6_SHAPE=64,128,128,8 python tools/ppo_phase_profile.py > $O/k6_phase_c4.txt 2>&1
ERL_K6_ARITH=f32 K6_SHAPE=64,128,128,8 python tools/ppo_phase_profile.py > $O/k6_phase_c4_f32.txt 2>&1
tools/bin/split_mfma_probe > $O/split_mfma_probe.txt 2>&1
K6_SHAPE=3,128,64,1 python tools/ppo_phase_profile.py > $O/k6_phase_c2.txt 2>&1
python tools/sac_fused_profile.py > $O/sac_fused_profile.txt 2>&1
for h in 128,128 256,128 256,128,64; do python tools/mlpn_step_profile.py $h; done > $O/mlpn_step_profile.txt 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# the bench command itself under the kernel trace (the judged command: python bench.py --gpus 1 --steps 20 --warmup 5)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_under_rocprof.json 2> /dev/null
cp $O/prof_c4/c4_kernel_stats.csv $O/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 > /dev/null 2>&1
cp $O/prof_c3/c3_kernel_stats.csv $O/c3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $O/prof_c2/c2_kernel_stats.csv $O/c2_kernel_stats.csv
# HBM traffic and SQ counters: one counter set per pass, kernel trace only (MI355X_MICROARCH.md)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
# (the fp32-MFMA minibatch kernel, for the comparison in DESIGN.md)
K6_ARITH=f32 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_f32 -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
K6_ARITH=f32 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_f32 -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
python tools/pmc_summarise.py $O/r03_pmc_traffic_f32.json $(find $O/pmc_fetch_f32 $O/pmc_write_f32 -name "*counter_collection.csv") > /dev/null 2>&1
python tools/pmc_summarise.py $O/r03_pmc_traffic.json $(find $O/pmc_fetch $O/pmc_write $O/pmc_sq -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
rm -rf $O/prof_c4/*trace* $O/prof_c3/*trace* $O/prof_c2/*trace* 2>/dev/null
tail -3 $O/pytest_gpu.log; cut -c1-200 $O/bench_c4.json; ls $O
