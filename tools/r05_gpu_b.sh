#!/bin/bash
# round 5, second call: the whole GPU suite; the bench line with the UNBRACKETED minibatch-kernel spans next to the bracketed ones; config 3
# with its in-loop roofline; the look-back tiling sweep; rocprofv3 kernel statistics of c4 / cd; counter passes for config 3's kernels and
# the replay gather by size.    gpurun -- bash tools/r05_gpu_b.sh [tag]
TAG=${1:-b}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
python tools/box_record.py > $O/box.json 2> $O/box.err
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err
python tools/gae_lb_sweep.py > $O/gae_lb_sweep.txt 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 0 > $O/bench_c4_under_rocprof.json 2> /dev/null
cp $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1) $O/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cd -o cd -- python bench.py --config cd --steps 3 --warmup 2 --no-cpu-baseline --no-smi --repeats 0 > $O/bench_cd_under_rocprof.json 2> /dev/null
cp $(find $O/prof_cd -name "*kernel_stats.csv" | head -1) $O/cd_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1) $O/c3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  d=$O/pmc_c3_$(echo $c | cut -d' ' -f1)
  C3_PART=step rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python tools/c3_pmc_workload.py > /dev/null 2>&1
done
PMC_KEEP_TEMPLATE=1 python tools/pmc_summarise.py $O/r05_c3_pmc_by_pass.json $(find $O/pmc_c3_* -name "*counter_collection.csv") > $O/pmc_c3_by_pass.txt 2>&1
python tools/pmc_summarise.py $O/r05_c3_pmc.json $(find $O/pmc_c3_* -name "*counter_collection.csv") > $O/pmc_c3.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  C3_PART=k9 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_k9_$c -o p -- python tools/c3_pmc_workload.py > $O/k9_cases_$c.txt 2>&1
done
PMC_CASES="replay_sample_kernel:5:seqs64_B256,seqs64_B4096,seqs64_B1048576,seqs1_B256,seqs1_B4096,seqs1_B1048576" python tools/pmc_summarise.py $O/r05_k9_pmc_by_size.json $(find $O/pmc_k9_* -name "*counter_collection.csv") > $O/pmc_k9.txt 2>&1
python tools/kstats_summarise.py $O/r05_kernel_times.json $O/c4_kernel_stats.csv "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 0" > $O/kernel_times.txt 2>&1
rm -rf $O/prof_c4 $O/prof_cd $O/prof_c3 $O/pmc_c3_* $O/pmc_k9_*
tail -4 $O/pytest_gpu.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).readline())
    except Exception as e:
        print(f.split('/')[-1], "unreadable", e, open(f.replace('.json', '.err')).read()[-600:] if 'rocprof' not in f else ''); continue
    r = d.get("roofline") or {}
    print(f.split('/')[-1], d.get("value"), d.get("ms_per_step"), (d.get("extra") or {}).get("repeated_regions_ms_per_step"), r.get("kernel"), r.get("avg_launch_us"), r.get("frac"),
          r.get("box_ratio"), r.get("shader_mhz"), r.get("bracketed"), r.get("phase_cycles"), {k: v for k, v in (d.get("breakdown") or {}).items() if k in ("slab_reduce_us", "clip_adam_us", "per_minibatch_rest_us", "boundaries_and_rest_per_minibatch_us", "update_net_ms", "explore_env_ms")})
    if "c3" in f:
        print("  c3 sample by size:", (d.get("roofline_sample") or {}).get("kernel_span_by_size"), d.get("us_per_update"))
    if "c4.json" in f:
        print("  gae sweep:", [(x["H"], x["N"], x.get("kernel_us"), x.get("call_us"), x.get("frac")) for x in d["roofline_gae"].get("sweep", [])])
b = json.load(open("$O/box.json"))
print("box k6:", b.get("k6_standalone"), b.get("hbm_copy_GBps"))
PY
cat $O/gae_lb_sweep.txt | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
from collections import defaultdict
by=defaultdict(list)
for r in rows: by[(r['H'],r['N'])].append(r)
for k,v in by.items():
    v.sort(key=lambda r:r['kernel_us']); print(k, [(r['algo'][:2], r['L'], r['W'], r['kernel_us']) for r in v[:5]], 'default', [r['kernel_us'] for r in v if r['algo']=='lookback' and r['L'] is None])
"
head -30 $O/pmc_k9.txt | head -5; python -c "
import json
d=json.load(open('$O/r05_k9_pmc_by_size.json'))['kernels']
for k,v in d.items():
    if 'replay' in k: print(k, v.get('hbm_read_bytes'), v.get('hbm_write_bytes'), v.get('avg_duration_us'))
d=json.load(open('$O/r05_c3_pmc_by_pass.json'))['kernels']
for k,v in d.items(): print(k, v.get('hbm_bytes_per_launch'), v.get('avg_duration_us'), v.get('mfma_busy_frac_of_kernel_time_at_2.4GHz'))
"
grep -E "ppo_step_s3|reduce_exch|clip_adam|rollout_fused|gae_" $O/c4_kernel_stats.csv $O/cd_kernel_stats.csv | cut -c1-200
