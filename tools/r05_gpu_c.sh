#!/bin/bash
# round 5, third call: the measurement hooks without atomics (primary region against the repeats), the single-launch tail A/B
# (ERL_FUSED_TAIL=0 | 3, alternating processes on one box), the batched look-back walk, the whole GPU suite.
TAG=${1:-c}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
python tools/box_record.py > $O/box.json 2> $O/box.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
for v in 0 3 0 3; do
  n=$(ls $O | grep -c "tail${v}_")
  ERL_FUSED_TAIL=$v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/tail${v}_$n.json 2> /dev/null
done
python tools/gae_lb_sweep.py > $O/gae_lb_sweep.txt 2>&1
python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
ERL_FUSED_TAIL=3 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5_tail3.json 2> /dev/null
python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
ERL_FUSED_TAIL=3 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2_tail3.json 2> /dev/null
tail -4 $O/pytest_gpu.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    if "box" in f: continue
    try:
        d = json.loads(open(f).readline())
    except Exception as e:
        print(f.split('/')[-1], "unreadable", e); continue
    r = d.get("roofline") or {}
    print(f.split('/')[-1], d.get("value"), d.get("ms_per_step"), (d.get("extra") or {}).get("repeated_regions_ms_per_step"), r.get("avg_launch_us"), r.get("frac"),
          r.get("box_ratio"), r.get("shader_mhz"), (r.get("bracketed") or {}).get("span_us"), {k: v for k, v in (d.get("breakdown") or {}).items() if k in ("slab_reduce_us", "clip_adam_us", "per_minibatch_rest_us", "boundaries_and_rest_per_minibatch_us", "update_net_ms", "explore_env_ms")})
    if "bench_c4.json" in f:
        print("  phases", r.get("phase_cycles")); print("  gae sweep:", [(x["H"], x["N"], x.get("kernel_us"), x.get("call_us"), x.get("frac")) for x in d["roofline_gae"].get("sweep", [])])
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
    v.sort(key=lambda r:r['kernel_us']); print(k, [(r['algo'][:2], r['L'], r['W'], r['kernel_us']) for r in v[:4]], 'default', [r['kernel_us'] for r in v if r['algo']=='lookback' and r['L'] is None], 'exact', [r['kernel_us'] for r in v if r['algo']=='exact'])
"
