// Sustained v_mfma_f32_32x32x16_f16 rate on MI355X: what the matrix cores deliver at the clocks the chip actually
// holds under this load (the 2.5 PFLOP/s figure assumes 2.4 GHz).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float seed, int rot) {
  h8 a, b;
  unsigned rng = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) {
    // seed 0: all-zero operands; otherwise pseudo-random values in [-2, 2) with full mantissa toggling
    rng = rng * 1664525u + 1013904223u; a[i] = (_Float16)(seed == 0.0f ? 0.0f : ((rng >> 8) & 0xffff) / 16384.0f - 2.0f);
    rng = rng * 1664525u + 1013904223u; b[i] = (_Float16)(seed == 0.0f ? 0.0f : ((rng >> 8) & 0xffff) / 16384.0f - 2.0f);
  }
  // "rot": 8 different A and B operand registers, a different pair for every MFMA (operand buses toggle like in a real GEMM);
  // otherwise the same pair every time
  h8 av[8], bv[8];
  for (int r = 0; r < 8; ++r)
    for (int i = 0; i < 8; ++i) {
      rng = rng * 1664525u + 1013904223u; av[r][i] = (_Float16)(seed == 0.0f ? 0.0f : ((rng >> 8) & 0xffff) / 16384.0f - 2.0f);
      rng = rng * 1664525u + 1013904223u; bv[r][i] = (_Float16)(seed == 0.0f ? 0.0f : ((rng >> 8) & 0xffff) / 16384.0f - 2.0f);
    }
  f16x c0 = {}, c1 = {}, c2 = {}, c3 = {};
  if (rot) {
    for (int it = 0; it < iters; it += 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0], bv[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1], bv[1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[2], bv[2], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[3], bv[3], c3, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[4], bv[4], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[5], bv[5], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[6], bv[6], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[7], bv[7], c3, 0, 0, 0);
    }
  } else {
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 1024 * 256 * 4 * sizeof(float));
  for (int wgs : {256, 1024}) for (float seed : {0.0f, 1.37f}) for (int rot : {0, 1}) {
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<wgs, 256>>>(d, 1000, seed, rot); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<wgs, 256>>>(d, iters, seed, rot); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)wgs * 4 * iters * 4 * 32768.0;
    printf("workgroups %4d (x4 waves) data %s operands %s: %.2f ms, %.0f TFLOP/s\n", wgs, seed == 0.0f ? "zeros " : "random", rot ? "rotating" : "fixed   ", ms, fl / ms / 1e9);
  }
  return 0;
}
