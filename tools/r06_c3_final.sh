#!/bin/bash
# config 3 on the final build: rocprofv3 kernel statistics and the counter passes of the fused SAC step
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1) $O/c3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  d=$O/pmc_c3_$(echo $c | cut -d' ' -f1)
  C3_PART=step rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python tools/c3_pmc_workload.py > /dev/null 2>&1
done
PMC_KEEP_TEMPLATE=1 python tools/pmc_summarise.py $O/r06_c3_pmc_by_pass.json $(find $O/pmc_c3_* -name "*counter_collection.csv") > $O/pmc_c3_by_pass.txt 2>&1
python tools/pmc_summarise.py $O/r06_c3_pmc.json $(find $O/pmc_c3_* -name "*counter_collection.csv") > $O/pmc_c3.txt 2>&1
rm -rf $O/prof_c3 $O/pmc_c3_*
grep -E "critic_tile|actor_fwd|actor_bwd|dw_table|clip_adam|alpha" $O/c3_kernel_stats.csv | cut -c1-160
python - <<PY
import json
d = json.load(open("$O/r06_c3_pmc_by_pass.json"))["kernels"]
for k, v in d.items():
    if "critic_tile" in k: print(k, v.get("avg_duration_us"), v.get("hbm_bytes_per_launch"), v.get("mfma_busy_frac_of_kernel_time_at_2.4GHz"))
PY
