cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
rm -rf gpurun_out/pmcw
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcw/$name -o $name -- $B > gpurun_out/pmcw_$name.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
run b SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_FLAT
python scripts/pmc_summary.py gpurun_out/pmcw gpurun_out/pmc_wait.json | grep "k_field16" | cut -c1-900
