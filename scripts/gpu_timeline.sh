cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_tl -o t -- python bench.py --steps 6 --warmup 2 "$@" > gpurun_out/bench_tl.log 2>&1
grep -a '"metric"' gpurun_out/bench_tl.log | cut -c1-260
python scripts/rocpd_timeline.py gpurun_out/prof_tl/t_results.db 45 > gpurun_out/timeline_eval.txt; cut -c1-150 gpurun_out/timeline_eval.txt
