#!/bin/bash
# Build deephisto_amd/libdeephisto_hip_<name>.so with extra compiler flags (A/B experiments; tooling only).
# usage: tools/build_variant.sh <name> [extra hipcc flags...]
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/dh_variant_$name; mkdir -p $T
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off"
hipcc $F "$@" -c $R/deephisto_amd/csrc/resnet_kernels.hip -o $T/resnet_kernels.o &
hipcc $F "$@" -c $R/deephisto_amd/csrc/tile_kernels.hip -o $T/tile_kernels.o &
wait
hipcc -shared -fPIC --offload-arch=gfx950 $T/resnet_kernels.o $T/tile_kernels.o -o $R/deephisto_amd/libdeephisto_hip_$name.so
echo built $R/deephisto_amd/libdeephisto_hip_$name.so
