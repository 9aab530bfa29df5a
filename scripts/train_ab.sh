# A/B of environment settings on the training step (bench.py --train, default + w4 start), two interleaved rounds:
#   bash scripts/train_ab.sh "DSN_WGRAD16=c" "DSN_WGRAD16=d"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do for cfg in "$@"; do for w in default w4; do
  env $cfg python bench.py --train --weights $w --steps 40 --warmup 8 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-28s %-8s' % ('$cfg', '$w'), 'train step %.3f ms' % d['ms_per_step'], 'loss %.6f' % d['config']['final_loss'])"
done; done; done
