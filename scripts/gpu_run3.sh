cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/gpu_parity_report.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/parity_report.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|passed|failed" | tee gpurun_out/gpu_tests_summary.log
