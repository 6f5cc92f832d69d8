#!/bin/bash
# Build libdemfi_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wno-unused-result -Wno-pass-failed -Werror=inline-asm -Werror=unused-value"
# experiment builds only (--ablation / --trace below) take extra flags from the environment; the product compile line is fixed
XFLAGS="$DEMFI_EXTRA_FLAGS"
# conv.hip = the dispatcher (demfi_conv2d); the convolution kernels are units of their own (round 6: they compile in parallel, 2 min -> 1 min)
CONV_UNITS="conv conv_general conv_c64 conv_narrow conv_sep conv_wstream"
SRCS_HIP="pointwise.hip"
for u in $CONV_UNITS; do SRCS_HIP="$SRCS_HIP $u.hip"; done
SRCS_CPP="abi.cpp"
[ -f metrics.hip ] && SRCS_HIP="$SRCS_HIP metrics.hip"
[ -f fgac_window.hip ] && SRCS_HIP="$SRCS_HIP fgac_window.hip"
[ -f resblock.hip ] && SRCS_HIP="$SRCS_HIP resblock.hip"
[ -f gru.hip ] && SRCS_HIP="$SRCS_HIP gru.hip"
[ -f viz.hip ] && SRCS_HIP="$SRCS_HIP viz.hip"
[ -f wsconv.hip ] && SRCS_HIP="$SRCS_HIP wsconv.hip"
[ -f ctx.cpp ] && SRCS_CPP="$SRCS_CPP ctx.cpp"
[ -f png_codec.cpp ] && SRCS_CPP="$SRCS_CPP png_codec.cpp"
# stale objects must never be linked: a failed compile has to fail the build
rm -f ./*.o libdemfi_hip.so
pids=()
objs=()
# conv.hip: no SLP vectorisation -- the auto-packed v_pk_add_f32 of the epilogues need v_mov shuffles around the accumulator
# registers (250 instead of 128 VALU in the 64->64 epilogue) and packed f32 VALU is slow beside MFMAs (MI355X_MICROARCH.md)
CONV_FLAGS="-fno-slp-vectorize"
for s in $SRCS_HIP; do
  o="${s%.hip}.o"; objs+=("$o")
  xf=""; case "$s" in conv*.hip|resblock.hip|gru.hip|wsconv.hip) xf="$CONV_FLAGS" ;; esac
  $HIPCC $FLAGS $xf -c "$s" -o "$o" & pids+=($!)
done
for s in $SRCS_CPP; do
  o="${s%.cpp}.o"; objs+=("$o")
  $HIPCC $FLAGS -x hip -c "$s" -o "$o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done      # 'wait PID' returns that job's status: set -e stops on the first failure
$HIPCC --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o libdemfi_hip.so -lz -lpthread
echo "built $(pwd)/libdemfi_hip.so"
# --ablation: second library with the ablation variants / experimental kernels (DEMFI_PERSIST_VARIANT, DEMFI_SEP_VARIANT,
# DEMFI_CONV_Z, ...); use it with DEMFI_HIP_LIB=$(pwd)/libdemfi_hip_abl.so.  Never loaded by default.
# --trace: third library whose persistent 64->64 kernels stamp s_memtime at their phase boundaries (tools/phase_trace.py)
if [ "$1" = "--trace" ]; then
  tp=()
  for u in $CONV_UNITS gru resblock; do
    $HIPCC $FLAGS $XFLAGS $CONV_FLAGS -DDEMFI_TRACE -c $u.hip -o ${u}_trace.o & tp+=($!)
  done
  for p in "${tp[@]}"; do wait "$p"; done
  trc=()
  for o in "${objs[@]}"; do
    case "$o" in conv*.o|resblock.o|gru.o) trc+=("${o%.o}_trace.o") ;; *) trc+=("$o") ;; esac
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC "${trc[@]}" -o libdemfi_hip_trace.so -lz -lpthread
  echo "built $(pwd)/libdemfi_hip_trace.so"
fi
if [ "$1" = "--ablation" ]; then
  ap=()
  for u in $CONV_UNITS; do
    $HIPCC $FLAGS $XFLAGS $CONV_FLAGS -DDEMFI_ABLATION -c $u.hip -o ${u}_abl.o & ap+=($!)
  done
  for p in "${ap[@]}"; do wait "$p"; done
  abl=()
  for o in "${objs[@]}"; do
    case "$o" in conv*.o) abl+=("${o%.o}_abl.o") ;; *) abl+=("$o") ;; esac
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC "${abl[@]}" -o libdemfi_hip_abl.so -lz -lpthread
  echo "built $(pwd)/libdemfi_hip_abl.so"
fi
# --asan: host-side AddressSanitizer + UBSan build: the three host translation units -- the plan builder / arena
# planner / op interpreter (ctx.cpp), the ABI glue (abi.cpp) and the PNG codec that parses untrusted bytes (png_codec.cpp) --
# instrumented, linked with the ordinary kernel objects.  Run with tools/asan_check.sh (preloads the sanitizer runtime under python).
if [ "$1" = "--asan" ]; then
  # pointer-overflow is off on purpose: the sizing pass of demfi_ctx_create lays the plan out on a NULL base (addresses == workspace offsets)
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize=pointer-overflow -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan -fno-sanitize-recover=undefined"
  aso=()
  for o in "${objs[@]}"; do
    case "$o" in
      ctx.o|abi.o|png_codec.o)
        $HIPCC --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wno-unused-result $SAN -x hip -c "${o%.o}.cpp" -o "${o%.o}_asan.o"
        aso+=("${o%.o}_asan.o") ;;
      *) aso+=("$o") ;;
    esac
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan "${aso[@]}" -o libdemfi_hip_asan.so -lz -lpthread
  echo "built $(pwd)/libdemfi_hip_asan.so"
fi
