#!/bin/bash
# the PMC passes of tools/r03_measure_all.sh alone (re-stamps profiles/r03_pmc_traffic.json after a kernel source change)
O=$GRAFT_REPO_ROOT/gpurun_out/r03_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -o p -- python tools/pmc_traffic.py > /dev/null 2>&1
python tools/pmc_summarise.py $O/r03_pmc_traffic.json $(find $O/pmc_fetch $O/pmc_write $O/pmc_sq -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_under_rocprof.json 2> /dev/null
cp $O/prof_c4/c4_kernel_stats.csv $O/c4_kernel_stats.csv; rm -rf $O/prof_c4/*trace* 2>/dev/null
python bench.py > $O/bench_c4.json 2> $O/bench_c4.err
ls $O
