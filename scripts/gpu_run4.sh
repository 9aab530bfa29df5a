cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tee gpurun_out/gpu_tests_summary.log
python -c "
import torch, dsnerf_amd
from dsnerf_amd import _lib, synth
c,f = synth.make_body(); x = synth.pose_body(c); dev=torch.device('cuda:0')
sd = synth.make_state_dict(); pk=_lib.PackedParams(dev).update({k: torch.from_numpy(v) for k,v in sd.items()})
sc=_lib.Scene(torch.from_numpy(c), torch.from_numpy(f), dev); sc.set_frame(pk, torch.from_numpy(x), torch.from_numpy(synth.make_poses()), 5)
print('nn stats', _lib.nn_stats(sc))
" 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1b -o r1b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof2.log 2>&1
grep -o '{"metric.*' gpurun_out/bench_prof2.log | cut -c1-1500
python scripts/rocpd_summary.py gpurun_out/prof_r1b/r1b_results.db | cut -c1-150
