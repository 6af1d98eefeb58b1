# every number quoted in DESIGN.md for round 2: run on the GPU box (gpurun -- bash tools/measure_all.sh), results under gpurun_out/final/
set -x
cd /root/repo
mkdir -p gpurun_out/final
K6_REPS=40 python tools/k6_ab.py > gpurun_out/final/k6_ab.txt 2>&1; K6_BETWEEN=both python tools/k6_ab.py >> gpurun_out/final/k6_ab.txt 2>&1
python tools/ppo_phase_profile.py > gpurun_out/final/k6_phase.txt 2>&1
python tools/tail_bench.py > gpurun_out/final/tail_bench.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mib tools/mfma_issue_bench.hip 2>/dev/null && timeout 120 /tmp/mib > gpurun_out/final/mfma_issue_bench.txt 2>&1
python bench.py > gpurun_out/final/bench_c4.json 2>gpurun_out/final/bench_c4.err
for c in c2 c3 c5; do python bench.py --config $c > gpurun_out/final/bench_$c.json 2>gpurun_out/final/bench_$c.err; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof4 -o bench -- python /root/repo/bench.py > /dev/null 2>&1
python /root/repo/tools/rocpd_stats.py $(find /tmp/prof4 -name "*.db" | head -1) > /root/repo/gpurun_out/final/c4_kernel_stats.csv
python /root/repo/tools/rocpd_stats.py --by-grid $(find /tmp/prof4 -name "*.db" | head -1) > /root/repo/gpurun_out/final/c4_kernel_stats_by_grid.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o bench -- python /root/repo/bench.py --config c3 --steps 7 --warmup 2 > /dev/null 2>&1
python /root/repo/tools/rocpd_stats.py $(find /tmp/prof3 -name "*.db" | head -1) > /root/repo/gpurun_out/final/c3_kernel_stats.csv
tail -1 /root/repo/gpurun_out/final/bench_c4.json | cut -c1-300
