# posed-mesh fine level: frame time and list / search kernel times by cell count (DSN_NN_FINE_TARGET)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for t in 0 24000 32000 52000 62000; do
  echo "== target $t"
  DSN_NN_FINE_TARGET=$t python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('  frame %.3f ms  alone %.3f' % (d['ms_per_step'], d['config'].get('ms_per_frame_alone', 0)))"
  rm -rf gpurun_out/cells_$t
  DSN_NN_FINE_TARGET=$t rocprofv3 --kernel-trace --stats -d gpurun_out/cells_$t -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 --no-roofline > /dev/null 2>&1
  python scripts/rocpd_summary.py gpurun_out/cells_$t/r_results.db | cut -c1-110 | grep -E "k_nns_search|k_grid_count|k_grid_fill|k_grid_super|k_sample_gg|k_nns_scatter|k_nns_scan"
  rm -rf gpurun_out/cells_$t
done
