# phase cycle counters of k_field16<forward> for every library variant built with -DF16_TIMING=1 (scripts/variants.sh t_<name> "-DF16_TIMING=1 ...")
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for so in dual-space-nerf_amd/variants/t_*.so; do
  echo "== $(basename $so .so)"
  DSNERF_LIB=$PWD/$so python scripts/timing_probe.py 2>/dev/null | tail -5
done
