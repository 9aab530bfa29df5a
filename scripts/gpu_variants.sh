cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for so in dual-space-nerf_amd/variants/*.so; do
  n=$(basename $so .so)
  DSNERF_LIB=$PWD/$so python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --pipeline 1 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); c = d['config']
print('$n', 'frame %.2f ms' % d['ms_per_step'], 'accurate-pass fraction %.4f' % c.get('accurate_pass_sample_fraction', 0))"
done
done
