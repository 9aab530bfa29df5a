set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4
nproc
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/gpu_tests.log; tail -30 gpurun_out/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1; tail -5 gpurun_out/bench1.log
