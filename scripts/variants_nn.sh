#!/bin/bash
# like variants.sh, for dsn_nn.hip (list build, cell-major search): variants_nn.sh name "flags" ...  -> dual-space-nerf_amd/variants/<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; P=$ROOT/dual-space-nerf_amd; mkdir -p $P/variants
python $P/build.py > /dev/null
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-result -Wno-inline-asm $flags \
      -c $P/csrc/dsn_nn.hip -o $P/variants/$name.o 2> $P/variants/$name.log && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $P/build/dsn_api.o $P/build/dsn_geom.o $P/variants/$name.o $P/build/dsn_field.o \
      $P/build/dsn_field16.o $P/build/dsn_train.o $P/build/dsn_image.o -o $P/variants/$name.so && echo "built $name" ) &
done
wait
rm -f $P/variants/*.o
