# round-3 measurement: tests, the driver's bench line, kernel traces (eval default / w4, train), PMC passes (separate runs, kernel trace only),
# the other bench lines kept under profiles/
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash scripts/gpu_round3.sh TAG'   -> gpurun_out/r03_TAG_*
TAG=${1:-a}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03_$TAG
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tee ${O}_tests.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -E "smoke ok|Error|error" | cut -c1-200 | tee ${O}_smoke.txt
timeout 900 python bench.py > ${O}_bench.log 2> ${O}_bench.err; tail -1 ${O}_bench.log > ${O}_bench.json; python - ${O}_bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d['config']
print('ms_per_step', round(d['ms_per_step'], 3), 'alone', round(c['ms_per_frame_alone'], 3), 'h2h', round(c.get('host_to_host_ms', 0), 2), 'h2h caller', round(c.get('host_to_host_ms_after_a_caller_torch_cpu_op', 0), 2))
for k, v in c.get('by_weights', {}).items(): print(' ', k, round(v['ms_per_frame'], 3), 'screen', v['density_screen'], 'stop', v['early_stop'])
print('train', round(d['train']['train_ms_per_step'], 3), d['train']['roofline']['frac'], 'w4', round(d.get('train_w4', {}).get('train_ms_per_step', 0), 3))
print('roofline', {k: d['roofline'][k] for k in ('frac', 'kernel_ms', 'rocprof_kernel_ms', 'traffic') if k in d['roofline']})
PY
timeout 300 python bench.py --train --steps 30 --warmup 5 2>/dev/null | tail -1 > ${O}_train_bench.json
timeout 300 python bench.py --train --weights w4 --steps 30 --warmup 5 2>/dev/null | tail -1 > ${O}_train_bench_w4.json
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1"   # profiles: one frame at a time, clean per-kernel durations
rm -rf gpurun_out/prof_r gpurun_out/prof_w4 gpurun_out/pmcr gpurun_out/prof_t gpurun_out/pmct
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r -o r -- $B > ${O}_bench_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_r/r_results.db > ${O}_kernel_trace.txt; cut -c1-150 ${O}_kernel_trace.txt | head -16
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_w4 -o r -- $B --weights w4 > ${O}_w4_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_w4/r_results.db > ${O}_w4_kernel_trace.txt
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcr/$name -o $name -- $B --no-roofline > gpurun_out/pmcr_$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT TCC_MISS TCC_REQ
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
run grbm GRBM_GUI_ACTIVE
python scripts/pmc_summary.py gpurun_out/pmcr ${O}_pmc.json | cut -c1-300 | head -8
T="python bench.py --train --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o t -- $T > ${O}_train_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_t/t_results.db > ${O}_train_kernel_trace.txt; cut -c1-150 ${O}_train_kernel_trace.txt | head -12
runt() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmct/$name -o $name -- python bench.py --train --steps 3 --warmup 2 > gpurun_out/pmct_$name.log 2>&1; }
runt fetch FETCH_SIZE
runt write WRITE_SIZE
runt sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES
runt grbm GRBM_GUI_ACTIVE
python scripts/pmc_summary.py gpurun_out/pmct ${O}_train_pmc.json | cut -c1-260 | head -8
timeout 600 python bench.py --strong --no-cpu-baseline 2>/dev/null | tail -1 > ${O}_strong.json
DSN_BENCH_FORCE_DIST=1 timeout 600 python bench.py --strong --no-cpu-baseline 2>/dev/null | tail -1 > ${O}_strong_rccl.json
DSN_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > ${O}_weak_rccl.json
timeout 900 python bench.py --strong --emulate-world 8 --steps 5 --warmup 2 2>/dev/null | tail -1 > ${O}_strong_emulated8.json
timeout 900 python bench.py --emulate-world 8 --steps 8 --warmup 3 2>/dev/null | tail -1 > ${O}_weak_emulated8.json
for wt in w2 w3 w4; do timeout 600 python bench.py --weights $wt --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > ${O}_$wt.json; done
for f in strong strong_rccl weak_rccl w2 w3 w4 strong_emulated8 weak_emulated8 train_bench train_bench_w4; do python -c "
import json; d = json.load(open('${O}_$f.json')); print('$f', round(d['ms_per_step'], 2), 'ms', {k: round(v, 4) for k, v in d['config'].items() if k in ('max_over_mean', 'predicted_strong_scaling_efficiency', 'predicted_weak_scaling_efficiency')})"; done
rm -rf gpurun_out/prof_r gpurun_out/prof_w4 gpurun_out/pmcr gpurun_out/prof_t gpurun_out/pmct
