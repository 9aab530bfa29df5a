cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -s 2>&1 | grep -vE "^\s*$" | tail -40 | cut -c1-600
