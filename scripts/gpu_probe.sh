cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/train_phase_probe.py 2>&1 | tail -2
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_p -o t -- python scripts/train_phase_probe.py > gpurun_out/probe.log 2>&1
tail -1 gpurun_out/probe.log
python scripts/rocpd_timeline.py gpurun_out/prof_p/t_results.db 60 > gpurun_out/timeline_probe.txt; awk '{ if ($4+0 > 30) print }' gpurun_out/timeline_probe.txt | cut -c1-150; tail -1 gpurun_out/timeline_probe.txt
