cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for so in dual-space-nerf_amd/variants/*.so; do
  n=$(basename $so .so)
  rm -rf gpurun_out/pv_$n
  DSNERF_LIB=$PWD/$so rocprofv3 --kernel-trace --stats -d gpurun_out/pv_$n -o v -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/pv_$n.log 2>&1
  echo "== $n $(grep -o '"ms_per_step[^,]*' gpurun_out/pv_$n.log)"
  python scripts/rocpd_summary.py gpurun_out/pv_$n/v_results.db | grep -E "k_warp|k_nn|k_normal" | cut -c1-110
done
