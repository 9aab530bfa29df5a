cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/gpu_debug_sampler.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/debug_sampler.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/gpu_tests.log; tail -60 gpurun_out/gpu_tests.log
