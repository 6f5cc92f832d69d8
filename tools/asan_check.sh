#!/bin/bash
# Host-side sanitizer run (VERDICT r5 missing #3): builds libdemfi_hip_asan.so (ctx.cpp, abi.cpp, png_codec.cpp under
# AddressSanitizer + UBSan) and runs, with it as the library,
#   * the CPU plan tests (plan builder, arena planner, descriptor builder, weight packer: tests/test_host.py) and the codec tests,
#   * a byte-flip / truncation fuzz of the PNG decoder (tools/png_fuzz.py): the only parser of untrusted bytes in the library.
# No GPU needed (nothing is launched: the contexts are bound on_host).  usage: tools/asan_check.sh [fuzz iterations, default 4000]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/demfi_amd/csrc"
[ -f conv.o ] && [ -f libdemfi_hip.so ] || bash build.sh > /dev/null
# the link step of build.sh removes nothing we need: re-use the kernel objects, instrument the host units
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# pointer-overflow is off on purpose: the sizing pass of demfi_ctx_create lays the plan out on a NULL base (addresses == workspace offsets)
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize=pointer-overflow -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan -fno-sanitize-recover=undefined"
objs=()
for o in *.o; do
  case "$o" in *_asan.o|*_trace.o|*_abl.o|ctx.o|abi.o|png_codec.o) ;; *) objs+=("$o") ;; esac
done
for u in ctx abi png_codec; do
  $HIPCC --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wno-unused-result $SAN -x hip -c $u.cpp -o ${u}_asan.o
  objs+=(${u}_asan.o)
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan "${objs[@]}" -o libdemfi_hip_asan.so -lz -lpthread
RT=$($HIPCC -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
export DEMFI_HIP_LIB=$ROOT/demfi_amd/csrc/libdemfi_hip_asan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== plan / packer / codec tests under ASan + UBSan"
LD_PRELOAD=$RT python -m pytest tests/test_host.py tests/test_clipio.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
echo "== PNG decoder fuzz under ASan + UBSan"
LD_PRELOAD=$RT python tools/png_fuzz.py ${1:-4000}
