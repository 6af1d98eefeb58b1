#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o m -- python tools/mlpn_step_profile.py 256,128 > $O/run.txt 2>&1
python - <<'PY'
import csv,collections,os
f=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r03k/prof/m_kernel_trace.csv")
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    name=r["Kernel_Name"][:60]; grid=(r.get("Grid_Size_X") or r.get("Grid_Size") , r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))
    agg[(name,grid)].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
tot=0
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    v=v[len(v)//3:]   # drop warm-up third
    print(f"{k[0]:60s} grid {k[1]} calls {len(v):4d} avg {sum(v)/len(v)/1e3:7.2f} us  total/iter {sum(v)/20/1e3:7.1f}")
PY
