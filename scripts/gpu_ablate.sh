cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cp dual-space-nerf_amd/libdsnerf_hip.so /tmp/real.so
for A in 3 4; do cp gpurun_ablate$A.so dual-space-nerf_amd/libdsnerf_hip.so || continue; echo "ablate $A:"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"kernel_ms[^,]*'; done
cp /tmp/real.so dual-space-nerf_amd/libdsnerf_hip.so
