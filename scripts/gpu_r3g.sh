# round 3, call G: the two field-kernel levers of VERDICT r02 #6 as timed variants inside one call (interleaved, two rounds):
#   base         the shipped kernels
#   mix0         -DF16_MIX=0: hi / lo split of the pipelined epilogues with plain VALU (cvt, sub, pack) instead of v_fma_mix*
#   noadd        -DF16_ABL=32 (WRONG results, timing only): the epilogue without the second accumulator's add / fold = what "one
#                accumulator per output block, two blocks' chains interleaved" would save
#   noadd_mix0   both
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2 3; do
for n in base mix0 noadd noadd_mix0; do
  DSNERF_LIB=$PWD/dual-space-nerf_amd/variants/$n.so python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; v = r.get('reverse_kernel', {})
print('$n', $rep, 'frame %.2f ms' % d['ms_per_step'], 'fwd %.3f ms / %d = %.4f us per k-sample' % (r['kernel_ms'], r['samples_per_launch'], 1e6 * r['kernel_ms'] / r['samples_per_launch']),
      'rev %.3f ms / %d = %.4f us per k-sample' % (v.get('kernel_ms', 0), v.get('samples_per_launch', 1), 1e6 * v.get('kernel_ms', 0) / max(v.get('samples_per_launch', 1), 1)))"
done
done | tee gpurun_out/r03g_field_levers_ab.txt
