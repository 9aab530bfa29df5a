# HBM traffic / MFMA occupancy of the training step's kernels (separate --pmc passes, kernel trace only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --train --steps 3 --warmup 2"
rm -rf gpurun_out/pmct
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmct/$name -o $name -- $B > gpurun_out/pmct_$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run grbm GRBM_GUI_ACTIVE
python scripts/pmc_summary.py gpurun_out/pmct gpurun_out/pmc_train.json | cut -c1-300 | head -14
