cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300
timeout 600 python bench.py --train --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-1200
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_t -o t -- python bench.py --train --steps 5 --warmup 2 > gpurun_out/bench_train_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_t/t_results.db > gpurun_out/kernel_trace_train.txt; cut -c1-150 gpurun_out/kernel_trace_train.txt | head -24
python scripts/rocpd_timeline.py gpurun_out/prof_t/t_results.db 200 > gpurun_out/timeline_train.txt; awk '{ if ($4+0 > 100) print }' gpurun_out/timeline_train.txt | cut -c1-150; tail -1 gpurun_out/timeline_train.txt
