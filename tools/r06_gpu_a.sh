#!/bin/bash
# round 6, first call: the split-residual probe, the whole GPU suite, config 4 with and without the one-instruction residual (alternating
# processes on ONE box), the GAE tiling sweep warm and cold.    gpurun -- bash tools/r06_gpu_a.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r06_a; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
tools/bin/dot2_split_probe > $O/dot2_split_probe.json 2> $O/dot2_split_probe.err; cat $O/dot2_split_probe.json
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
for rep in 0 1; do
  for v in main nodot; do
    lib=$L/liberl_hip.so; [ $v != main ] && lib=$L/liberl_hip_$v.so
    [ -f $lib ] || continue
    ERL_HIP_LIB=$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/${v}_$rep.json 2> $O/${v}_$rep.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*_?.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "mhz", r["shader_mhz"], "fetch", r.get("instruction_fetch"), "reduce", b["slab_reduce_us"], "adam", b["clip_adam_us"],
              "phases", r["phase_cycles"])
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python tools/gae_lb_sweep.py 200x4096 1024x4096 2048x4096 32x4096 > $O/gae_lb_sweep.txt 2>&1
cat $O/gae_lb_sweep.txt | cut -c1-200
