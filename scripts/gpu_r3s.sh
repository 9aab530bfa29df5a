cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_abi.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2 3; do for v in 1 0; do
DSN_BENCH_SHARE_CUS=$v timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/r03s_$v.json
python -c "
import json
d=json.load(open('gpurun_out/r03s_$v.json')); c=d['config']; print('share $v', round(d['ms_per_step'],3), 'alone', round(c['ms_per_frame_alone'],3), {k: round(x['ms_per_frame'],2) for k,x in c['by_weights'].items()}, 'h2h', round(c['host_to_host_ms'],2))"
done; done
