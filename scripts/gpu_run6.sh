cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tee gpurun_out/gpu_tests_summary.log
python scripts/gpu_parity_report.py 2>&1 | grep -E "field|==" | head -12
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '{"metric.*' | cut -c1-1800
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fp32 2>&1 | grep -o '"roofline.*' | cut -c1-400
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1d -o r1d -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_prof4.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_r1d/r1d_results.db | cut -c1-150 | head -8
