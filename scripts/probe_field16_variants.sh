#!/bin/bash
# compile-only probe: register spills of k_field16 for the variant switches (no GPU needed)
cd "$(dirname "$0")/../dual-space-nerf_amd/csrc"
for S in 0 1; do for R in 0 1; do for F in 0 1; do for P in 0 1; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DF16_SINCOS_OCML=$S -DF16_RELU_ASM=$R -DF16_FENCE=$F -DF16_PREFETCH=$P \
      -Rpass-analysis=kernel-resource-usage -c dsn_field16.hip -o /tmp/probe_${S}${R}${F}${P}.o > /tmp/probe_${S}${R}${F}${P}.log 2>&1
    echo "sincos_ocml=$S relu_asm=$R fence=$F prefetch=$P : $(grep -A12 'Name: _Z9k_field16' /tmp/probe_${S}${R}${F}${P}.log | grep -E 'Scratch|SGPRs Spill|VGPRs Spill' | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' ')" ) &
done; done; wait; done; done
