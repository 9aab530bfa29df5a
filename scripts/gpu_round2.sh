# round-2 measurement: tests, the driver's bench line, kernel traces (eval + train), PMC passes (separate runs, kernel trace only)
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash scripts/gpu_round2.sh TAG'   -> gpurun_out/r02_TAG_*
TAG=${1:-a}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r02_$TAG
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tee ${O}_tests.txt
timeout 900 python bench.py > ${O}_bench.log 2> ${O}_bench.err; tail -1 ${O}_bench.log > ${O}_bench.json; cut -c1-600 ${O}_bench.json
timeout 300 python bench.py --train --steps 30 --warmup 5 2>/dev/null | tail -1 > ${O}_train_bench.json; cut -c1-400 ${O}_train_bench.json
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1"   # profiles: one frame at a time, clean per-kernel durations
rm -rf gpurun_out/prof_r gpurun_out/pmcr gpurun_out/prof_t gpurun_out/pmct
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r -o r -- $B > ${O}_bench_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_r/r_results.db > ${O}_kernel_trace.txt; cut -c1-150 ${O}_kernel_trace.txt | head -16
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcr/$name -o $name -- $B --no-roofline > gpurun_out/pmcr_$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT TCC_MISS TCC_REQ
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
run grbm GRBM_GUI_ACTIVE
python scripts/pmc_summary.py gpurun_out/pmcr ${O}_pmc.json | cut -c1-300 | head -8
T="python bench.py --train --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o t -- $T > ${O}_train_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_t/t_results.db > ${O}_train_kernel_trace.txt; cut -c1-150 ${O}_train_kernel_trace.txt | head -24
# the other bench lines kept under profiles/: strong scaling frame (configs[3]) alone and through RCCL with one rank, the weak line through RCCL,
# the trained-like parameter sets
timeout 600 python bench.py --strong --no-cpu-baseline 2>/dev/null | tail -1 > ${O}_strong.json
DSN_BENCH_FORCE_DIST=1 timeout 600 python bench.py --strong --no-cpu-baseline 2>/dev/null | tail -1 > ${O}_strong_rccl.json
DSN_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > ${O}_weak_rccl.json
for wt in w2 w3; do timeout 600 python bench.py --weights $wt --no-cpu-baseline 2>/dev/null | tail -1 > ${O}_$wt.json; done
for f in strong strong_rccl weak_rccl w2 w3; do python -c "
import json; d = json.load(open('${O}_$f.json')); print('$f', round(d['ms_per_step'], 2), 'ms')"; done
