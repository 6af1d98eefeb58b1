#!/bin/bash
# one more box of the pool: the box record (clock probe, the minibatch kernel stand-alone / cold / in the update loop, per-workgroup
# placement) and the config-4 bench line.  ~40 s.   gpurun -- bash tools/r05_box_sample.sh <tag>
TAG=${1:-s}
O=$GRAFT_REPO_ROOT/gpurun_out/r05_box_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
python tools/box_record.py > $O/box.json 2> $O/box.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gae-sweep > $O/bench_c4.json 2> $O/bench_c4.err
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
if [ -f $L/liberl_hip_e16.so ]; then
  for n in "" 0 1; do
    if [ -z "$n" ]; then ERL_HIP_LIB=$L/liberl_hip_e16.so python tools/k6_only_net.py >> $O/k6_only_net.jsonl 2>/dev/null
    else ERL_HIP_LIB=$L/liberl_hip_e16.so ERL_K6_ONLY_NET=$n python tools/k6_only_net.py >> $O/k6_only_net.jsonl 2>/dev/null; fi
  done
fi
[ -n "$2" ] && eval "$2"
python - <<PY
import json
d = json.loads(open("$O/bench_c4.json").readline())
r = d["roofline"]
print("c4", d["value"], d["ms_per_step"], d["extra"]["repeated_regions_ms_per_step"], "k6", r["avg_launch_us"], "ratio", r["box_ratio"], "mhz", r["shader_mhz"])
print("  phases", r["phase_cycles"])
w = d["clocks"]["k6_workgroups_last_sampled_launch"]
print("  wg:", {k: w[k] for k in ("distinct_cus", "workgroups_in_a_second_round", "start_us", "dur_us", "end_us_max", "dur_us_mean_by_xcc")}, w["late_starters"][:4])
b = json.load(open("$O/box.json"))
for k, v in b["k6_standalone"].items():
    ww = v.get("workgroups") or {}
    print("box", k, v.get("us_back_to_back_events"), v.get("us_span_unbracketed"), v.get("workgroup_us"), v.get("shader_mhz"), ww.get("dur_us"), ww.get("dur_us_mean_by_xcc"), ww.get("workgroups_in_a_second_round"))
print(b["clock_probe"].get("k6_like_forward_mix"), b["clock_probe"].get("code_walk_56KB"), b.get("hbm_copy_GBps"))
import os
if os.path.exists("$O/k6_only_net.jsonl"):
    for ln in open("$O/k6_only_net.jsonl"):
        r = json.loads(ln); print("only_net", r["only_net"], r["us_span"], r["workgroup_us"], r["shader_mhz"], r["dur_us"])
PY
