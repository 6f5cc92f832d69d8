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
