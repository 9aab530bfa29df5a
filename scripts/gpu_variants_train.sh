# A/B of library variants on the training step (bench.py --train) inside one GPU call
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for so in dual-space-nerf_amd/variants/*.so; do
  n=$(basename $so .so)
  DSNERF_LIB=$PWD/$so python bench.py --train --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$n', 'train step %.2f ms' % d['ms_per_step'])"
done
done
