#!/bin/bash
# round 5, first call: what box is this, are the tests green at ABI 17, the driver's bench line with clocks / phases / box_ratio,
# the reference-default shape (--config cd), and what the phase stamps cost the unsampled launches (A/B against the no-stamp build).
#   gpurun -- bash tools/r05_gpu_a.sh [tag]
TAG=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
export ERL_QUIET=1
python tools/box_record.py > $O/box.json 2> $O/box.err
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
for v in nostamp main nostamp main; do
  lib=$L/liberl_hip.so; [ $v = nostamp ] && lib=$L/liberl_hip_nostamp.so
  [ -f $lib ] || continue
  n=$(ls $O | grep -c "ab_$v")
  ERL_HIP_LIB=$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/ab_${v}_$n.json 2> /dev/null
done
python bench.py --config cd --gpus 1 --steps 5 --warmup 2 --repeats 2 > $O/bench_cd.json 2> $O/bench_cd.err
K6_LOOP=1 K6_SHAPE=64,128,128,8 python tools/ppo_phase_profile.py > $O/k6_phase_c4.txt 2>&1
tail -3 $O/pytest_gpu.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f.split('/')[-1], "unreadable", e); continue
    if "box.json" in f:
        print("box", d.get("host"), d.get("clock_probe"), d.get("k6_standalone"), d.get("hbm_copy_GBps"), d.get("partition"))
    else:
        r = d.get("roofline", {})
        print(f.split('/')[-1], d.get("value"), d.get("ms_per_step"), (d.get("extra") or {}).get("repeated_regions_ms_per_step"), r.get("kernel"), r.get("avg_launch_us"),
              r.get("box_ratio"), r.get("shader_mhz"), r.get("phase_cycles"), (d.get("roofline_gae") or {}).get("frac"), (d.get("breakdown") or {}).get("per_minibatch_rest_us"))
PY
grep -A14 "actor: total" $O/k6_phase_c4.txt | head -20
