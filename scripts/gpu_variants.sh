cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for so in dual-space-nerf_amd/variants/*.so; do
  n=$(basename $so .so)
  DSNERF_LIB=$PWD/$so python bench.py --steps 8 --warmup 3 --no-cpu-baseline --pipeline 1 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$n', 'frame %.2f ms' % d['ms_per_step'], 'screen %.2f' % r.get('screen_kernel', {}).get('kernel_ms', 0), 'fwd %.2f' % r['kernel_ms'], 'rev %.2f' % r.get('reverse_kernel', {}).get('kernel_ms', 0))"
done
done
