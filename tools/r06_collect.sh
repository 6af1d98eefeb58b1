#!/bin/bash
# copies the summaries of a closing call (tools/r06_final.sh <tag>) from gpurun_out/r06_<tag>/ into profiles/ under the names DESIGN.md / README.md cite
O=gpurun_out/r06_${1:-final}
for c in c2 c3 c5 cw cd c1; do cp $O/bench_$c.json profiles/bench_r06_$c.json; done
cp $O/bench_c4.json profiles/bench_r06_c4_closing_call.json
cp $O/bench_c4_under_rocprof.json profiles/bench_r06_c4_under_rocprof.json
cp $O/bench_cd_under_rocprof.json profiles/bench_r06_cd_under_rocprof.json
for c in c4 cd c3; do cp $O/${c}_kernel_stats.csv profiles/r06_${c}_kernel_stats.csv; done
cp $O/r06_pmc_traffic.json $O/r06_c3_pmc.json $O/r06_c3_pmc_by_pass.json $O/r06_kernel_times.json profiles/
cp $O/box.json profiles/r06_box.json
cp $O/gae_lb_sweep.txt profiles/r06_gae_lb_sweep_final.txt
tail -4 $O/pytest_gpu.log > profiles/r06_pytest_gpu_tail.txt
cp $O/k6_phase_c4.txt profiles/r06_k6_phase_c4.txt
cp $O/k6_phase_c2.txt profiles/r06_k6_phase_c2.txt
python tools/kernel_code_sizes.py > profiles/r06_kernel_code_sizes.txt 2>&1
