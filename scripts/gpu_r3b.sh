# round 3, call B: GPU tests incl. the w4 cases; the driver's bench line (by_weights, train, host path); strong-scaling emulation;
# overlap analysis of the pipelined frame; one-frame-at-a-time kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03b
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-400 | tee ${O}_tests.txt
timeout 900 python bench.py > ${O}_bench.log 2> ${O}_bench.err; tail -1 ${O}_bench.log > ${O}_bench.json; python - <<'PY'
import json
d = json.load(open('gpurun_out/r03b_bench.json'))
c = d['config']
print('ms_per_step', round(d['ms_per_step'], 3), 'alone', round(c['ms_per_frame_alone'], 3), 'h2h', round(c.get('host_to_host_ms', 0), 2), 'h2h caller', round(c.get('host_to_host_ms_after_a_caller_torch_cpu_op', 0), 2), 'threads', c.get('host_threads'), 'quota', c.get('host_cpu_quota_cores'))
for k, v in c.get('by_weights', {}).items(): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
print('train', {k: v for k, v in d.get('train', {}).items() if k in ('train_ms_per_step', 'value')}, d.get('train', {}).get('roofline', {}).get('frac'))
print('roofline', {k: d['roofline'][k] for k in ('frac', 'kernel_ms', 'rocprof_kernel_ms', 'traffic') if k in d['roofline']})
print('cpu', d.get('cpu_baseline'))
PY
timeout 900 python bench.py --strong --emulate-world 8 --steps 5 --warmup 2 2>/dev/null | tail -1 > ${O}_strong_emulated8.json; python - <<'PY'
import json
d = json.load(open('gpurun_out/r03b_strong_emulated8.json'))['config']
print('shares', [round(s['ms'], 2) for s in d['shares']], 'max/mean', round(d['max_over_mean'], 4), 'whole', round(d['whole_frame_one_gpu_ms'], 2), 'eff', round(d['predicted_strong_scaling_efficiency'], 3), 'undeal', round(d['undeal_scatter_ms'], 3))
PY
rm -rf gpurun_out/prof_ov gpurun_out/prof_r
rocprofv3 --kernel-trace -d gpurun_out/prof_ov -o t -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > ${O}_ov.log 2>&1
python scripts/rocpd_overlap.py gpurun_out/prof_ov/t_results.db 4 > ${O}_overlap.txt; cut -c1-160 ${O}_overlap.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 > ${O}_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_r/r_results.db > ${O}_kernel_trace.txt; cut -c1-150 ${O}_kernel_trace.txt | head -24
rm -rf gpurun_out/prof_ov gpurun_out/prof_r
