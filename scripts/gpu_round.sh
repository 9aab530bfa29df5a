# full round-end measurement: tests, bench line, kernel trace, PMC passes (separate runs, no sys traces)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300
timeout 600 python bench.py > gpurun_out/bench_round.log 2>&1; tail -1 gpurun_out/bench_round.log | cut -c1-2500
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --pipeline 1"   # profiles: one frame at a time, clean per-kernel durations
rm -rf gpurun_out/prof_r gpurun_out/pmcr
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r -o r -- $B > gpurun_out/bench_prof_r.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_r/r_results.db > gpurun_out/kernel_trace_r.txt; cut -c1-150 gpurun_out/kernel_trace_r.txt | head -24
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcr/$name -o $name -- $B --no-roofline > gpurun_out/pmcr_$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT TCC_MISS TCC_REQ
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
run grbm GRBM_GUI_ACTIVE
python scripts/pmc_summary.py gpurun_out/pmcr gpurun_out/pmc_round.json | cut -c1-400 | head -12
