cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do for p in 2 3; do for k in 10 20; do
timeout 300 python bench.py --pipeline $p --steps $k --warmup 2 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 > gpurun_out/r03r_p$p.json
python -c "
import json
d=json.load(open('gpurun_out/r03r_p$p.json')); print('pipeline $p steps $k', round(d['ms_per_step'],3))"
done; done; done
