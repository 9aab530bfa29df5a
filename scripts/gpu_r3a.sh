# round 3, call A: GPU tests after the ABI-3 / lazy-colour / fine-only / tangent-guard edits, the driver's bench line, host-pool probe, w4 training
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tee gpurun_out/r03a_tests.txt
timeout 900 python bench.py > gpurun_out/r03a_bench.log 2> gpurun_out/r03a_bench.err; tail -1 gpurun_out/r03a_bench.log > gpurun_out/r03a_bench.json; cut -c1-700 gpurun_out/r03a_bench.json
timeout 300 python scripts/h2h_guard_probe.py 2>/dev/null | tail -1 > gpurun_out/r03a_h2h_guard.json; python -c "
import json; d=json.load(open('gpurun_out/r03a_h2h_guard.json')); print(d['info'])
for r in d['runs']: print(r['host_pool_limit'], r['caller'], round(r['render_view_ms'],2), round(r['loop_ms'],2), r['throttled_ms'], r['nr_throttled'])"
timeout 1300 python scripts/train_w4.py --steps 24000 > gpurun_out/w4_train.log 2>&1; grep -v amdgpu.ids gpurun_out/w4_train.log | cut -c1-400 | tail -32
