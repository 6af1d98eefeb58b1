#!/bin/bash
# K6 with order 2 as the default: (a) is the kernel's end the drain of its slab stores? (nost: dW2 tiles not stored -- timing only), (b) tile boundaries
# inside the second layer and the backward (proffine), (c) order 0 vs 2 once more on another box
O=$GRAFT_REPO_ROOT/gpurun_out/r06_k; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
for rep in 0 1; do
  for a in dwo0 main nost; do
    lib=$L/liberl_hip.so; [ $a != main ] && lib=$L/liberl_hip_$a.so
    ERL_HIP_LIB=$lib timeout 300 python bench.py --config c4 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep --no-smi --repeats 3 > $O/c4_${a}_$rep.json 2> $O/c4_${a}_$rep.err
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c*_*_?.json")):
    try:
        d = json.loads(open(f).readline()); r = d["roofline"]; b = d["breakdown"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "update_ms", b["update_net_ms"], {k: v for k, v in b.items() if "us" in k}, "phases", r.get("phase_cycles"))
    except Exception as e:
        print(f, "FAILED", e)
PY
for a in prof proffine; do
  ERL_HIP_PROF_LIB=$L/liberl_hip_$a.so K6_LOOP=1 python tools/ppo_phase_profile.py > $O/phase_$a.txt 2>&1
  grep "fine stamps" $O/phase_$a.txt | cut -c1-700
  grep -A14 "actor: total" $O/phase_$a.txt | cut -c1-120
done
