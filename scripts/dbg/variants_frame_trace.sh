#!/bin/bash
# kernel-trace lines matching PATTERN of one frame at a time, per library variant: variants_frame_trace.sh PATTERN name ...   ("-" = the in-tree library)
PAT=$1; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "$@"; do
  L=""; [ "$v" != "-" ] && L="DSNERF_LIB=dual-space-nerf_amd/variants/$v.so"
  rm -rf gpurun_out/pv_$v
  env $L rocprofv3 --kernel-trace --stats -d gpurun_out/pv_$v -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 --no-roofline > /dev/null 2>&1
  echo "== $v"; python scripts/rocpd_summary.py gpurun_out/pv_$v/r_results.db 80 | grep -E "$PAT" | cut -c1-130
  rm -rf gpurun_out/pv_$v
done
