cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tee gpurun_out/gpu_tests_summary.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench_full.log 2>&1; grep -o '{"metric.*' gpurun_out/bench_full.log | cut -c1-2500
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1e -o r1e -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_prof5.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_r1e/r1e_results.db | cut -c1-150 | head -14
