cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03l
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-400 | tee ${O}_tests.txt
B="python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-roofline"
for rep in 1 2; do for wt in default w4; do
  timeout 300 $B --weights $wt 2>/dev/null | tail -1 > ${O}_${wt}_$rep.json
  python -c "
import json
d=json.load(open('${O}_${wt}_$rep.json')); print('$wt', $rep, round(d['ms_per_step'],3), 'alone', round(d['config']['ms_per_frame_alone'],3))"
done; done
rm -rf gpurun_out/prof_r
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 > ${O}_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/prof_r/r_results.db > ${O}_kernel_trace.txt; cut -c1-150 ${O}_kernel_trace.txt | head -16
rm -rf gpurun_out/prof_r
