#!/bin/bash
# Plain-C host of the C ABI (tests/c/forward_golden.c): gcc + the HIP runtime C API + libdemfi_hip.so
set -e
cd "$(dirname "$0")"
ROOT=../..
gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I$ROOT/include -I/opt/rocm/include forward_golden.c -o forward_golden \
    -L$ROOT/demfi_amd/csrc -ldemfi_hip -L/opt/rocm/lib -lamdhip64 -lm \
    -Wl,-rpath,'$ORIGIN/../../demfi_amd/csrc' -Wl,-rpath,/opt/rocm/lib
echo "built $(pwd)/forward_golden"
