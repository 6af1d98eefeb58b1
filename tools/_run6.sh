cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02g; mkdir -p $O
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err
python tools/rocpd_stats.py $O/prof/bench_results.db --by-grid > $O/bench_kernel_stats_by_grid.csv
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/r02g/bench_kernel_stats_by_grid.csv')) if 'gae' in r['Name']]
w=csv.DictWriter(open('gpurun_out/r02g/gae_kernel_stats_by_size.csv','w'),fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows: print(r['Name'][:40], r['Grid'], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
rm -rf $O/prof
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
