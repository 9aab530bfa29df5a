#!/bin/bash
# build variants of the TRAINING kernels (dsn_train.hip + dsn_field16.hip with extra -D flags, always -DDSN_EXPERIMENTS) into
# dual-space-nerf_amd/variants/<name>.so; usage: variants_train.sh name "flags" [name "flags" ...]   (DSNERF_LIB selects one at run time)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; P=$ROOT/dual-space-nerf_amd; mkdir -p $P/variants
python $P/build.py > /dev/null
CF="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-result -Wno-inline-asm -DDSN_EXPERIMENTS"
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc $CF $flags -c $P/csrc/dsn_field16.hip -o $P/variants/$name.f16.o 2> $P/variants/$name.log && \
    /opt/rocm/bin/hipcc $CF $flags -c $P/csrc/dsn_train.hip -o $P/variants/$name.train.o 2>> $P/variants/$name.log && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $P/build/dsn_api.o $P/build/dsn_geom.o $P/build/dsn_nn.o $P/build/dsn_field.o \
      $P/variants/$name.f16.o $P/variants/$name.train.o $P/build/dsn_image.o -o $P/variants/$name.so && echo "built $name" ) &
done
wait
rm -f $P/variants/*.o
