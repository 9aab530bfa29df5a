cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tee gpurun_out/gpu_tests_summary.log
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z_0-9]*(MFMA|WAIT|BUSY_CY|WAVE_CYCLES|ACTIVE_INST|INSTS_VALU|INST_CYCLES|WAVES)[A-Z_0-9]*|TCC_(HIT|MISS|REQ|EA0_RDREQ)[A-Z_0-9_]*|TCP_[A-Z_]*(HIT|MISS|TCC_READ)[A-Z_0-9]*|FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|MfmaUtil|VALUBusy)\b" | sort -u | tr '\n' ' ' > gpurun_out/pmc_names.txt; cat gpurun_out/pmc_names.txt | cut -c1-3000
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1c -o r1c -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof3.log 2>&1
grep -o '{"metric.*' gpurun_out/bench_prof3.log | cut -c1-1800
python scripts/rocpd_summary.py gpurun_out/prof_r1c/r1c_results.db | cut -c1-150 | head -14
