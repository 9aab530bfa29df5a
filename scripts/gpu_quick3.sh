cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300
python scripts/gpu_parity_report.py 2>&1 | grep -E "shade" | head -6
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_q -o q -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_q.log 2>&1
grep -o '"ms_per_step[^,]*' gpurun_out/bench_q.log
python scripts/rocpd_summary.py gpurun_out/prof_q/q_results.db | cut -c1-150 | head -8
