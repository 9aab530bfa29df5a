# per-kernel times of the training step for every library variant (rocprofv3 kernel trace of bench.py --train), inside one GPU call
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for so in dual-space-nerf_amd/variants/*.so; do
  n=$(basename $so .so); rm -rf gpurun_out/vt_$n
  DSNERF_LIB=$PWD/$so rocprofv3 --kernel-trace --stats -d gpurun_out/vt_$n -o t -- python bench.py --train --weights default --steps 5 --warmup 2 > /dev/null 2>&1
  echo "== $n"; python scripts/rocpd_summary.py gpurun_out/vt_$n/t_results.db | cut -c1-120 | grep -E "k_field16|k_tangent16|k_adjoint16|k_t_wgrad16" ; rm -rf gpurun_out/vt_$n
done
