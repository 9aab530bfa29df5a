cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-roofline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-40s' % '$*', 'frame %.3f ms' % d['ms_per_step'], 'alone %.3f' % d['config'].get('ms_per_frame_alone', 0))"; }
for rep in 1 2; do
run --pipeline 2
run --pipeline 3
run --pipeline 4
DSN_BENCH_SHARE_CUS=0 run --pipeline 3
DSN_PERSISTENT_GROUPS=240 DSN_BENCH_SHARE_CUS=0 run --pipeline 3
done
