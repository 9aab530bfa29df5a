# A/B of environment switches inside one GPU call: bash scripts/gpu_ab_env.sh "NAME=VAL ..." "NAME=VAL ..." ...   (each argument = one configuration)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%-40s' % '$cfg', 'frame %.2f ms' % d['ms_per_step'], 'screen %.2f' % r.get('screen_kernel', {}).get('kernel_ms', 0), 'fwd %.2f' % r['kernel_ms'], 'rev %.2f' % r.get('reverse_kernel', {}).get('kernel_ms', 0))"
done
done
