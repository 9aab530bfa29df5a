# round 3, call D: kernel traces of the converged set (w4: one frame at a time), training kernel trace + PMC passes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03d
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tee ${O}_tests.txt
rm -rf gpurun_out/prof_w4 gpurun_out/prof_t gpurun_out/pmct
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_w4 -o r -- python bench.py --weights w4 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --pipeline 1 > ${O}_w4_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_w4/r_results.db > ${O}_w4_kernel_trace.txt; cut -c1-150 ${O}_w4_kernel_trace.txt | head -40
python scripts/rocpd_timeline.py gpurun_out/prof_w4/r_results.db 14 > ${O}_w4_timeline.txt
T="python bench.py --train --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o t -- $T > ${O}_train_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_t/t_results.db > ${O}_train_kernel_trace.txt; cut -c1-150 ${O}_train_kernel_trace.txt | head -40
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmct/$name -o $name -- python bench.py --train --steps 3 --warmup 2 > gpurun_out/pmct_$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES
run grbm GRBM_GUI_ACTIVE
python scripts/pmc_summary.py gpurun_out/pmct ${O}_train_pmc.json | cut -c1-330 | head -24
rm -rf gpurun_out/prof_w4 gpurun_out/prof_t gpurun_out/pmct
