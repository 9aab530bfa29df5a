cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do for o in new old; do
DSN_RV_ORDER=$o timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/r03t.json
python -c "
import json
d=json.load(open('gpurun_out/r03t.json')); c=d['config']; print('$o', round(d['ms_per_step'],3), 'alone', round(c['ms_per_frame_alone'],3), 'h2h', round(c['host_to_host_ms'],2), round(c['host_to_host_ms_after_a_caller_torch_cpu_op'],2), 'chunked', round(c['chunked_frame']['host_to_host_ms'],2))"
done; done
