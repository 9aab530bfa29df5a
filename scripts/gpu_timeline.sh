cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_tl -o t -- python bench.py --train --steps 4 --warmup 2 > gpurun_out/bench_tl.log 2>&1
tail -1 gpurun_out/bench_tl.log | cut -c1-300
python scripts/rocpd_timeline.py gpurun_out/prof_tl/t_results.db 75 > gpurun_out/timeline_train.txt; tail -150 gpurun_out/timeline_train.txt | cut -c1-150
