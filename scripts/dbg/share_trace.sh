#!/bin/bash
# kernel trace of the emulated 8-rank strong partition, one share at a time (--pipeline 1): where a share's time goes beyond its
# eighth of the frame.  -> gpurun_out/${TAG}_share_kernel_trace.txt (totals over all shares), ${TAG}_share_sequence.txt (the last share)
TAG=${1:-x}; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/profs_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/profs_$TAG -o s -- python bench.py --strong --emulate-world 8 --emulate-sweep 8 --steps ${STEPS:-4} --warmup 2 --pipeline ${PIPE:-1} --no-cpu-baseline > gpurun_out/${TAG}_share_trace.log 2>&1
python scripts/rocpd_summary.py gpurun_out/profs_$TAG/s_results.db 60 > gpurun_out/${TAG}_share_kernel_trace.txt
python scripts/rocpd_sequence.py gpurun_out/profs_$TAG/s_results.db k_pose_setup > gpurun_out/${TAG}_share_sequence.txt
python scripts/rocpd_period.py gpurun_out/profs_$TAG/s_results.db k_pose_setup | tail -40 > gpurun_out/${TAG}_share_period.txt
rm -rf gpurun_out/profs_$TAG
tail -1 gpurun_out/${TAG}_share_trace.log | cut -c1-300
