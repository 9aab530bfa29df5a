cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for p in 3 4 2; do
timeout 300 python bench.py --pipeline $p --steps 24 --warmup 4 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 > gpurun_out/r03s_p.json
python -c "
import json
d=json.load(open('gpurun_out/r03s_p.json')); c=d['config']; print('pipeline $p', round(d['ms_per_step'],3))"
done; done
