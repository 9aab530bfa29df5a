cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-400
