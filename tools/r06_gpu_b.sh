#!/bin/bash
# round 6, second call: the whole GPU suite (dot2 off again), GAE sweep warm / cold-behind-writes / cold-behind-reads with plain and
# non-temporal input loads.    gpurun -- bash tools/r06_gpu_b.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r06_b; mkdir -p $O
cd $GRAFT_REPO_ROOT
export ERL_QUIET=1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
python tools/gae_lb_sweep.py 200x4096 1024x4096 2048x4096 > $O/gae_lb_sweep.txt 2>&1
ERL_HIP_LIB=$L/liberl_hip_gaent.so python tools/gae_lb_sweep.py 200x4096 1024x4096 2048x4096 > $O/gae_lb_sweep_nt_loads.txt 2>&1
grep -h '"L": null' $O/gae_lb_sweep.txt $O/gae_lb_sweep_nt_loads.txt | grep lookback
python bench.py --config cd --no-cpu-baseline --no-smi --repeats 0 > $O/bench_cd.json 2> $O/bench_cd.err
ERL_HIP_LIB=$L/liberl_hip_gaent.so python bench.py --config cd --no-cpu-baseline --no-smi --repeats 0 > $O/bench_cd_nt.json 2> $O/bench_cd_nt.err
python - <<PY
import json
for f in ("bench_cd", "bench_cd_nt"):
    try:
        d = json.loads(open("$O/%s.json" % f).readline()); print(f, d["value"], d["ms_per_step"], json.dumps(d.get("roofline_gae"))[:600])
    except Exception as e:
        print(f, "FAILED", e)
PY
