#!/bin/bash
# dev helper: recompile the named units (default: gru) and relink libdemfi_hip.so (+ the trace library with gru_trace.o)
set -e
cd /root/repo/demfi_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wno-unused-result -Wno-pass-failed -Werror=inline-asm -Werror=unused-value"
for u in ${@:-gru}; do
  case $u in
    gru|conv|conv_*|resblock|wsconv) /opt/rocm/bin/hipcc $F -fno-slp-vectorize -c $u.hip -o $u.o ;;
    ctx|abi|png_codec) /opt/rocm/bin/hipcc $F -x hip -c $u.cpp -o $u.o ;;
    *) /opt/rocm/bin/hipcc $F -c $u.hip -o $u.o ;;
  esac
done
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -DDEMFI_TRACE -c gru.hip -o gru_trace.o
rm -f *.hipfb
O="conv.o conv_general.o conv_c64.o conv_narrow.o conv_sep.o conv_wstream.o wsconv.o pointwise.o metrics.o fgac_window.o resblock.o viz.o abi.o ctx.o png_codec.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O gru.o -o libdemfi_hip.so -lz -lpthread
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O gru_trace.o -o libdemfi_hip_trace.so -lz -lpthread
echo relinked
