# round 3, call E: the training step on listed rows - gradient tests, exactness vs the dense evaluation, step time A/B, kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03e
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-400 | tee ${O}_tests.txt
for rep in 1 2; do
  DSN_TRAIN_ALL_ROWS=1 timeout 300 python bench.py --train --steps 30 --warmup 5 2>/dev/null | tail -1 > ${O}_train_dense_$rep.json
  timeout 300 python bench.py --train --steps 30 --warmup 5 2>/dev/null | tail -1 > ${O}_train_rows_$rep.json
  python -c "
import json
for k in ('dense', 'rows'):
    d = json.load(open('${O}_train_' + k + '_$rep.json')); print(k, $rep, round(d['ms_per_step'], 3), 'ms', round(d['roofline']['frac'], 3), d['config'].get('final_loss'))"
done
rm -rf gpurun_out/prof_t
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o t -- python bench.py --train --steps 5 --warmup 2 > ${O}_train_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_t/t_results.db > ${O}_train_kernel_trace.txt; cut -c1-150 ${O}_train_kernel_trace.txt | head -40
rm -rf gpurun_out/prof_t
