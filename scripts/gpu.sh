#!/bin/bash
# ONE parametrised runner for everything that is measured on the GPU box (replaces round 3's gpu_r3a..t.sh one-offs):
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash scripts/gpu.sh TAG step [step ...]'      -> gpurun_out/TAG_*
# steps (run in the order given):
#   tests            the -m gpu suite (summary lines)            tests:EXPR      only tests matching -k EXPR
#   smoke            __graft_entry__.smoke()
#   bench            the driver's line: python bench.py           bench:ARGS      python bench.py ARGS (commas = spaces), e.g. bench:--weights,default,--no-extras
#   train            bench.py --train (default + w4 start)
#   trace[:ARGS]     rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 ARGS`
#   pmc[:ARGS]       the separate --pmc passes of the same command (FETCH_SIZE, WRITE_SIZE, TCC, SQ, GRBM) -> TAG_pmc*.json
#   poison           pytest -m gpu with every scratch allocation of the binding filled with 0xFF first (DSN_POISON_SCRATCH=1)
#   traintrace / trainpmc   the same two for `bench.py --train --steps 5 --warmup 2`
#   strong[:ARGS] / weak8 / strong8[:ARGS]    bench.py --strong (one GPU); --emulate-world 8 in the weak mode; the strong partition emulated at 2 / 4 / 8 ranks
#   dist1            the RCCL path with ONE rank (DSN_BENCH_FORCE_DIST=1) in weak, --strong and --train mode
#   world2           TWO ranks sharing this one GPU over gloo (debug): the real multi-rank code paths of all three modes end to end
#   ab:"A=1 B=2":"C=3"   A/B of environment settings (two interleaved rounds), BENCH_ARGS from the environment
#   py:SCRIPT[:ARGS] python scripts/SCRIPT ARGS
TAG=${1:-x}; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/$TAG
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1"
summ() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d.get('config', {})
print('ms_per_step', round(d['ms_per_step'], 3), 'value', round(d['value']), 'n_gpus', d['n_gpus'], 'alone', c.get('ms_per_frame_alone'), 'weights', c.get('weights'), 'early', (c.get('early_stop') or {}).get('enabled'), 'screen', c.get('density_screen'))
for k, v in (c.get('by_weights') or {}).items(): print('  ', k, round(v['ms_per_frame'], 3), 'screen', v['density_screen'], 'stop', v['early_stop'])
for k in ('train', 'train_w4'):
    if k in d: print(' ', k, round(d[k]['train_ms_per_step'], 3), {a: round(b, 4) for a, b in d[k]['roofline'].items() if isinstance(b, float)})
if 'roofline' in d: print('  roofline', {k: d['roofline'][k] for k in ('kernel', 'frac', 'kernel_ms', 'samples_per_launch', 'traffic') if k in d['roofline']})
if 'ranks' in d: print('  ranks', d['ranks'])
PY
}
pmcpass() { dir=$1; shift; name=$1; shift; cmd=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $dir/$name -o $name -- $cmd > ${dir}_$name.log 2>&1; }
for step in "$@"; do
  arg=""; case "$step" in *:*) arg="${step#*:}"; step="${step%%:*}";; esac
  args="${arg//,/ }"
  echo "== $step $args"
  case "$step" in
    tests) if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$args" 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-400 | tee ${O}_tests_k.txt
           else timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-400 | tee ${O}_tests.txt; fi;;
    poison) DSN_POISON_SCRATCH=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-400 | tee ${O}_tests_poisoned.txt;;
    smoke) timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -E "smoke ok|Error|error" | cut -c1-200 | tee ${O}_smoke.txt;;
    bench) n=$(echo "bench$arg" | tr -c 'a-zA-Z0-9\n' '_'); timeout 1200 python bench.py $args > ${O}_$n.log 2> ${O}_$n.err; tail -1 ${O}_$n.log > ${O}_$n.json; summ ${O}_$n.json || tail -5 ${O}_$n.err;;
    train) for w in default w4; do timeout 400 python bench.py --train --weights $w --steps 30 --warmup 5 $args 2>/dev/null | tail -1 > ${O}_train_$w.json; summ ${O}_train_$w.json; done;;
    trace) rm -rf gpurun_out/prof_$TAG; rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o r -- $B --no-roofline $args > ${O}_trace.log 2>&1
           python scripts/rocpd_sequence.py gpurun_out/prof_$TAG/r_results.db k_pose_setup > ${O}_frame_sequence.txt; python scripts/rocpd_period.py gpurun_out/prof_$TAG/r_results.db k_pose_setup > ${O}_frame_period.txt
           python scripts/rocpd_summary.py gpurun_out/prof_$TAG/r_results.db > ${O}_kernel_trace$(echo "$arg" | tr -c 'a-zA-Z0-9\n' '_').txt; cut -c1-160 ${O}_kernel_trace$(echo "$arg" | tr -c 'a-zA-Z0-9\n' '_').txt | head -${TRACE_LINES:-24}; rm -rf gpurun_out/prof_$TAG;;
    pmc) D=gpurun_out/pmc_$TAG; rm -rf $D; C="$B --no-roofline $args"
         pmcpass $D fetch "$C" FETCH_SIZE; pmcpass $D write "$C" WRITE_SIZE; pmcpass $D tcc "$C" TCC_HIT TCC_MISS TCC_REQ
         pmcpass $D sq "$C" SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
         pmcpass $D grbm "$C" GRBM_GUI_ACTIVE
         python scripts/pmc_summary.py $D ${O}_pmc$(echo "$arg" | tr -c 'a-zA-Z0-9\n' '_').json | cut -c1-300 | head -10; rm -rf $D;;
    traintrace) rm -rf gpurun_out/proft_$TAG; rocprofv3 --kernel-trace --stats -d gpurun_out/proft_$TAG -o t -- python bench.py --train --steps 5 --warmup 2 $args > ${O}_traintrace.log 2>&1
           python scripts/rocpd_summary.py gpurun_out/proft_$TAG/t_results.db > ${O}_train_kernel_trace.txt; python scripts/rocpd_sequence.py gpurun_out/proft_$TAG/t_results.db k_sample_gg > ${O}_train_sequence.txt; python scripts/rocpd_period.py gpurun_out/proft_$TAG/t_results.db > ${O}_train_period.txt; cut -c1-160 ${O}_train_kernel_trace.txt | head -${TRACE_LINES:-24}; rm -rf gpurun_out/proft_$TAG;;
    trainpmc) D=gpurun_out/pmct_$TAG; rm -rf $D; C="python bench.py --train --steps 3 --warmup 2 $args"
         pmcpass $D fetch "$C" FETCH_SIZE; pmcpass $D write "$C" WRITE_SIZE
         pmcpass $D sq "$C" SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES; pmcpass $D grbm "$C" GRBM_GUI_ACTIVE
         python scripts/pmc_summary.py $D ${O}_train_pmc.json 5 | cut -c1-300 | head -12; rm -rf $D;;
    strong) timeout 600 python bench.py --strong --no-cpu-baseline $args 2>/dev/null | tail -1 > ${O}_strong$(echo "$arg" | tr -c 'a-zA-Z0-9\n' '_').json; summ ${O}_strong$(echo "$arg" | tr -c 'a-zA-Z0-9\n' '_').json;;
    strong8) # one frame partitioned for N emulated ranks, each share rendered alone on this GPU: strong8 = the metric's 512 x 512 x 64 frame at
             # 2 / 4 / 8 ranks in one run; strong8:--big-frame = configs[3]; strong8:--partition,tiles = the round-robin deal
             n=$(echo "strong_emulated$arg" | tr -c 'a-zA-Z0-9\n' '_')
             timeout 1500 python bench.py --strong --emulate-world 8 --emulate-sweep ${SWEEP:-2,4,8} --steps ${EMU_STEPS:-12} --warmup 3 $args 2>${O}_$n.err | tail -1 > ${O}_$n.json
             python - ${O}_$n.json <<'PY' || tail -5 ${O}_$n.err
import json, sys
d = json.load(open(sys.argv[1])); c = d['config']
w = c['whole_frame_one_gpu']; print('whole frame: %.3f ms pipelined, %.3f alone' % (w['ms'], w['ms_alone']))
for n, v in sorted(c['worlds'].items(), key=lambda kv: int(kv[0])):
    print(' N=%s  share max %.3f mean %.3f (alone max %.3f; host enqueue %.3f)  max/mean %.3f  sum/whole %.3f  -> %.3f ms/frame, speed-up %.2f (eff %.2f; one at a time %.2f)  %s' % (
        n, v['share_ms_max'], v['share_ms_mean'], v['share_ms_alone_max'], v.get('host_enqueue_ms_max', 0), v['max_over_mean'], v['sum_of_shares_over_whole_frame'], v['predicted_ms_per_frame'],
        v['predicted_speedup'], v['predicted_strong_scaling_efficiency'], v['predicted_speedup_one_frame_at_a_time'], v['partition'].get('bounds', v['partition'].get('tile_rays'))))
PY
             ;;
    weak8) timeout 900 python bench.py --emulate-world 8 --steps 8 --warmup 3 $args 2>/dev/null | tail -1 > ${O}_weak_emulated8.json; summ ${O}_weak_emulated8.json;;
    dist1) DSN_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > ${O}_weak_rccl.json; summ ${O}_weak_rccl.json
           DSN_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --strong --no-cpu-baseline 2>/dev/null | tail -1 > ${O}_strong_rccl.json; summ ${O}_strong_rccl.json
           DSN_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --train --steps 10 --warmup 3 2>/dev/null | tail -1 > ${O}_train_rccl.json; summ ${O}_train_rccl.json;;
    world2) # the REAL multi-rank code paths with 2 ranks on this one GPU over gloo (RCCL refuses two ranks per device): control flow, collectives and
            # reassembly of all three modes end to end; the ranks share the GPU, so the times mean nothing
            for m in "" "--weak" "--partition tiles" "--train"; do
              n=$(echo "world2$m" | tr -c 'a-zA-Z0-9\n' '_')
              DSN_BENCH_BACKEND=gloo DSN_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline $m > ${O}_$n.log 2> ${O}_$n.err
              tail -1 ${O}_$n.log > ${O}_$n.json; summ ${O}_$n.json || tail -5 ${O}_$n.err; done;;
    ab) IFS=':' read -ra CFGS <<< "$arg"
        for rep in 1 2; do for cfg in "${CFGS[@]}"; do
          env $cfg python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-roofline $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-44s' % '$cfg', 'frame %.3f ms' % d['ms_per_step'], 'alone %.3f' % d['config'].get('ms_per_frame_alone', 0))" | tee -a ${O}_ab.txt
        done; done;;
    py) s="${arg%%:*}"; a=""; case "$arg" in *:*) a="${arg#*:}";; esac; timeout 1500 python scripts/$s ${a//,/ } 2>&1 | tail -${PY_LINES:-40} | tee ${O}_$(basename $s .py).txt;;
    *) echo "unknown step $step";;
  esac
done
