# LDS / issue counters of the field kernels next to the same counters of the inner-loop micro-benchmark (scripts/ubench/field_loop.hip)
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash scripts/gpu_pmc_lds.sh'   -> gpurun_out/pmc_lds.json, pmc_lds_ubench.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 --no-roofline"
hipcc --offload-arch=gfx950 -O3 -o /tmp/fl scripts/ubench/field_loop.hip 2>/dev/null
rm -rf gpurun_out/pmcl gpurun_out/pmclu
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcl/$name -o $name -- $B > gpurun_out/pmcl_$name.log 2>&1
        rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmclu/$name -o $name -- /tmp/fl > gpurun_out/pmclu_$name.log 2>&1; }
run lds1 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run lds2 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
run iss1 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES
run iss2 SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run grbm GRBM_GUI_ACTIVE
python scripts/pmc_summary.py gpurun_out/pmcl gpurun_out/pmc_lds.json > /dev/null
python scripts/pmc_summary.py gpurun_out/pmclu gpurun_out/pmc_lds_ubench.json > /dev/null
ls gpurun_out/pmcl/*/ | head; tail -3 gpurun_out/pmcl_lds1.log
