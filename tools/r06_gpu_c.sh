#!/bin/bash
# round 6, third call: the interleaved replay ring -- its tests, config 3, the sample kernel's counter passes by size on both layouts.
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sac.py tests/test_per.py tests/test_agent_gpu.py tests/test_discrete_gpu.py -m gpu -q -x -k "replay or ring or sac or per or golden or lookback or gae" > $O/pytest_ring.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ring.log
tail -5 $O/pytest_ring.log
python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python - <<PY
import json
d = json.loads(open("$O/bench_c3.json").readline())
print("c3", d["value"], d["us_per_update"], json.dumps(d["roofline_sample"])[:1500])
PY
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ring in interleaved planar; do
  for c in FETCH_SIZE WRITE_SIZE; do
    C3_RING=$ring C3_PART=k9 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_k9_${ring}_$c -o p -- python tools/c3_pmc_workload.py > $O/k9_cases_${ring}_$c.txt 2>&1
  done
done
PMC_CASES="replay_sample_rows_kernel:5:seqs64_B256,seqs64_B4096,seqs64_B1048576,seqs1_B256,seqs1_B4096,seqs1_B1048576" python tools/pmc_summarise.py $O/r06_k9_pmc_by_size.json $(find $O/pmc_k9_interleaved_* -name "*counter_collection.csv") > $O/pmc_k9.txt 2>&1
PMC_CASES="replay_sample_kernel:5:seqs64_B256,seqs64_B4096,seqs64_B1048576,seqs1_B256,seqs1_B4096,seqs1_B1048576" python tools/pmc_summarise.py $O/r06_k9_pmc_by_size_planar.json $(find $O/pmc_k9_planar_* -name "*counter_collection.csv") > $O/pmc_k9_planar.txt 2>&1
rm -rf $O/pmc_k9_*
python - <<PY
import json
for f in ("r06_k9_pmc_by_size.json", "r06_k9_pmc_by_size_planar.json"):
    try:
        d = json.load(open("$O/" + f))["kernels"]
        for k, v in d.items():
            if "replay_sample" in k: print(f, k, v.get("avg_duration_us"), v.get("hbm_read_bytes"), v.get("hbm_write_bytes"))
    except Exception as e:
        print(f, "FAILED", e)
PY
