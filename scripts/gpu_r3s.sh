cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do for v in 0 1; do
if [ $v = 1 ]; then export DSN_FULL_ON_KEPT=1; else unset DSN_FULL_ON_KEPT; fi
timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 > gpurun_out/r03s_$v.json
python -c "
import json
d=json.load(open('gpurun_out/r03s_$v.json')); print('full_on_kept $v', round(d['ms_per_step'],3), 'alone', round(d['config']['ms_per_frame_alone'],3))"
done; done
