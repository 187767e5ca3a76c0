#!/bin/bash
# Builds liblqcd_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
ARCH=${LQCD_ARCH:-gfx950}
FLAGS="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
mkdir -p build
pids=()
for f in stencil fields blas ops capi; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ lqcd_internal.h -nt build/$f.o ] || [ ../../include/lqcd_hip.h -nt build/$f.o ]; then
    ( hipcc $FLAGS -c $f.hip -o build/$f.o ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=$ARCH -shared -fPIC -o liblqcd_hip.so build/stencil.o build/fields.o build/blas.o build/ops.o build/capi.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $(pwd)/liblqcd_hip.so"
