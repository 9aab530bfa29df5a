cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stages.py -m gpu -q -k "field" 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step[^,]*\|"roofline.*' | cut -c1-420
