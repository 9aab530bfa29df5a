cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --eager-baseline 2>&1 | tail -4 | cut -c1-1500 | tee gpurun_out/eager_baseline.log
timeout 600 python bench.py --train --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-1200 | tee gpurun_out/train_bench.log
