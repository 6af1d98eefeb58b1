cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02c; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/pytest.log; cat $O/pytest.log
timeout 300 python bench.py --no-gae-sweep --cpu-iters 3 > $O/bench.json 2> $O/bench.err; cat $O/bench.json
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -o p -- python tools/pmc_traffic.py > $O/pmc_$n.log 2>&1
done
python tools/pmc_summarise.py $O/pmc_traffic.json $(find $O -name "*counter_collection.csv") > $O/pmc_sum.log 2>&1; tail -5 $O/pmc_sum.log
find $O -name "*.csv" -size +2M -delete
