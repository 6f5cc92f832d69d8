#!/bin/bash
# Build libdemfi_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wno-unused-result"
$HIPCC $FLAGS -c conv.hip -o conv.o &
$HIPCC $FLAGS -c pointwise.hip -o pointwise.o &
$HIPCC $FLAGS -x hip -c abi.cpp -o abi.o &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC conv.o pointwise.o abi.o -o libdemfi_hip.so
echo "built $(pwd)/libdemfi_hip.so"
# --ablation: second library with the ablation variants / experimental kernels (DEMFI_PERSIST_VARIANT, DEMFI_SEP_VARIANT,
# DEMFI_CONV_Z, ...); use it with DEMFI_HIP_LIB=$(pwd)/libdemfi_hip_abl.so.  Never loaded by default.
if [ "$1" = "--ablation" ]; then
  $HIPCC $FLAGS -DDEMFI_ABLATION -c conv.hip -o conv_abl.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC conv_abl.o pointwise.o abi.o -o libdemfi_hip_abl.so
  echo "built $(pwd)/libdemfi_hip_abl.so"
fi
