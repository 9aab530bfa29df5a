cd $GRAFT_REPO_ROOT
for sw in "5 2" "30 5" "100 5"; do set -- $sw
python bench.py --train --steps $1 --warmup $2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); t=d.get('train') or {}; print('$sw', 'step %.3f ms' % d['ms_per_step'], d['config'].get('final_loss'), {k: d['roofline'].get(k) for k in ('forward_rows','backward_rows')})"
done
