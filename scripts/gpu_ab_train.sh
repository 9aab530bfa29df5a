#!/bin/bash
# A/B of environment settings on the training step, interleaved: bash scripts/gpu_ab_train.sh TAG "ENV=a" "ENV=b" ...   ("-" = no setting)
TAG=$1; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do for cfg in "$@"; do for w in default w4; do
  e="$cfg"; [ "$cfg" = "-" ] && e="DSN_NONE=1"
  env $e python bench.py --train --weights $w --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-50s %-8s step %.3f ms  loss %.6f' % ('$cfg', '$w', d['ms_per_step'], d['config']['final_loss']))" | tee -a gpurun_out/${TAG}_train_ab.txt
done; done; done
