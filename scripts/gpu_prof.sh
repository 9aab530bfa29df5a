cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1 -o r1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
tail -2 gpurun_out/bench_prof.log
find gpurun_out/prof_r1 -name "*stats*" | head; 
for f in $(find gpurun_out/prof_r1 -name "*kernel_stats.csv"); do head -20 $f; done
