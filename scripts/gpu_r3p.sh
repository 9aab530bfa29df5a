cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for L in 8 4 5 6 10 12 16 8; do
for wt in w4 w3; do
DSN_STOP_SLICE=$L timeout 300 python bench.py --weights $wt --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 > gpurun_out/r03p_${wt}_$L.json
python -c "
import json
d=json.load(open('gpurun_out/r03p_${wt}_$L.json')); c=d['config']; print('$wt L=$L', round(d['ms_per_step'],3), 'alone', round(c['ms_per_frame_alone'],3), c.get('early_stop'))" | cut -c1-260
done; done
