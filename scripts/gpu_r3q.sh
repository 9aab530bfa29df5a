cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_render.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2; do for wt in w4 w3 w2; do
timeout 300 python bench.py --weights $wt --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 > gpurun_out/r03q_${wt}.json
python -c "
import json
d=json.load(open('gpurun_out/r03q_${wt}.json')); c=d['config']; print('$wt', round(d['ms_per_step'],3), 'alone', round(c['ms_per_frame_alone'],3), c['early_stop']['skipped_fraction_of_non_transparent'])"
done; done
