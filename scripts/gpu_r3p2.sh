cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider -k "early_stop or converged or chunks or phase" 2>&1 | tail -3
for L in 2 3 4 8; do
for wt in w4 w3 w2; do
DSN_STOP_SLICE=$L timeout 300 python bench.py --weights $wt --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 > gpurun_out/r03p_${wt}_$L.json
python -c "
import json
d=json.load(open('gpurun_out/r03p_${wt}_$L.json')); c=d['config']; print('$wt L=$L', round(d['ms_per_step'],3), 'alone', round(c['ms_per_frame_alone'],3))"
done; done
