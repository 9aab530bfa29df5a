# round 3, call C: tests after the w4 fixes + phase split; A/B of the two overlap modes inside one call; overlap analysis of the phase mode
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03c
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-400 | tee ${O}_tests.txt
B="python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-roofline"
for rep in 1 2; do for ov in frame phase; do
  timeout 300 $B --overlap $ov 2>/dev/null | tail -1 > ${O}_ab_${ov}_$rep.json
  python -c "
import json; d=json.load(open('${O}_ab_${ov}_$rep.json')); print('$ov', $rep, round(d['ms_per_step'],3), 'alone', round(d['config']['ms_per_frame_alone'],3))"
done; done
for wt in w4 w2; do for ov in frame phase; do
  timeout 300 $B --overlap $ov --weights $wt 2>/dev/null | tail -1 > ${O}_ab_${wt}_${ov}.json
  python -c "
import json; d=json.load(open('${O}_ab_${wt}_${ov}.json')); print('$wt', '$ov', round(d['ms_per_step'],3), 'alone', round(d['config']['ms_per_frame_alone'],3))"
done; done
rm -rf gpurun_out/prof_ov
rocprofv3 --kernel-trace -d gpurun_out/prof_ov -o t -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-roofline --overlap phase > ${O}_ov.log 2>&1
python scripts/rocpd_overlap.py gpurun_out/prof_ov/t_results.db 4 > ${O}_overlap_phase.txt; cut -c1-160 ${O}_overlap_phase.txt
python scripts/rocpd_timeline.py gpurun_out/prof_ov/t_results.db 36 > ${O}_timeline_phase.txt
rm -rf gpurun_out/prof_ov
