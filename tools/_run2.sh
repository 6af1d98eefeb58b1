cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02b
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "ppo_step or update_loop or clip_adam" 2>&1 | tail -15) > gpurun_out/r02b/pytest_k6.log
cat gpurun_out/r02b/pytest_k6.log
for i in 1 2; do ERL_K6_FORM=8 timeout 120 python tools/k6_ab.py; timeout 120 python tools/k6_ab.py; done 2>&1 | tee gpurun_out/r02b/k6_ab.log
timeout 120 python tools/ppo_phase_profile.py 2>&1 | tee gpurun_out/r02b/phase_w4.log
