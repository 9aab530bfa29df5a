#!/bin/bash
# which training-backward kernel is the co-residency aggressor?  variants tuN = every training matrix kernel guarded (DSN_OWN_SIMD)
# EXCEPT those of the bits of N (scripts/variants_train.sh; bits: 1 k_tangent16, 2 k_adjoint16, 4 k_t_wgrad16d, 8 k_t_wgrad16p, 16 k_t_lin,
# 32 k_t_wgrad).  Victims: 14 frames' shading / geometry phases beside the backward of an 8192 x 64 step.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "$@"; do
  echo "== variant $v" | tee -a gpurun_out/race_bisect.txt
  DSNERF_LIB=$GRAFT_REPO_ROOT/dual-space-nerf_amd/variants/$v.so RACE_AGG=BACKWARD RACE_QUICK=1 timeout 300 python scripts/dbg/race_train.py ${RACE_REPS:-3} 2>&1 | grep -A2 "^aggressor" | tee -a gpurun_out/race_bisect.txt
done
