#!/bin/bash
# per library variant (dual-space-nerf_amd/variants/*.so): kernel trace of 5 bench frames, the lines of the kernels matching $1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; PAT=${1:-k_nns_search}
for so in dual-space-nerf_amd/variants/*.so; do
  n=$(basename $so .so); rm -rf gpurun_out/vt_$n
  DSNERF_LIB=$PWD/$so rocprofv3 --kernel-trace --stats -d gpurun_out/vt_$n -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 --no-roofline $BENCH_ARGS > /dev/null 2>&1
  echo "== $n"; python scripts/rocpd_summary.py gpurun_out/vt_$n/r_results.db | grep -E "$PAT" | cut -c1-130; rm -rf gpurun_out/vt_$n
done
