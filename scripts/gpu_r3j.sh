cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03j
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_round3.py tests/test_gpu_stages.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-400 | tee ${O}_tests.txt
for wt in default w4; do
timeout 300 python bench.py --train --weights $wt --steps 30 --warmup 5 2>/dev/null | tail -1 > ${O}_train_$wt.json
python -c "
import json
d=json.load(open('${O}_train_$wt.json')); print('$wt', {k: d[k] for k in d if 'ms' in k or 'rows' in k})"
done
DSN_TRAIN_FAR_SEARCH_MIN=99999999999 timeout 300 python bench.py --train --steps 30 --warmup 5 2>/dev/null | tail -1 > ${O}_train_nopre.json
python -c "
import json
d=json.load(open('${O}_train_nopre.json')); print('nopre', {k: d[k] for k in d if 'ms' in k})"
rm -rf gpurun_out/prof_t
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o t -- python bench.py --train --steps 5 --warmup 2 > ${O}_train_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_t/t_results.db > ${O}_train_kernel_trace.txt; cut -c1-150 ${O}_train_kernel_trace.txt | head -40
rm -rf gpurun_out/prof_t
