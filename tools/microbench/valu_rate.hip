// Issue cost (cycles per wave-instruction, one wave per SIMD, independent chains) of the VALU / LDS instructions the convolution
// epilogues are made of, on MI355X.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define BODY(name, txt)                                                                                         \
  __global__ __launch_bounds__(256, 1) void name(unsigned long long* out, int iters) {                          \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;                                   \
    __shared__ float lds[4096];                                                                                 \
    unsigned la = threadIdx.x * 16;                                                                             \
    lds[threadIdx.x] = a0;                                                                                      \
    __syncthreads();                                                                                            \
    unsigned long long t0 = __builtin_readcyclecounter();                                                       \
    for (int i = 0; i < iters; ++i) { asm volatile(txt : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), \
                                                        "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(la) : "memory"); } \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
    unsigned long long t1 = __builtin_readcyclecounter();                                                       \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                  \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345u) out[1] = (unsigned long long)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
  }
// 8 independent instructions per asm statement
BODY(k_cvt_pk, "v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %9, %10\n v_cvt_pk_f16_f32 %2, %10, %11\n v_cvt_pk_f16_f32 %3, %11, %12\n"
               "v_cvt_pk_f16_f32 %4, %12, %13\n v_cvt_pk_f16_f32 %5, %13, %14\n v_cvt_pk_f16_f32 %6, %14, %15\n v_cvt_pk_f16_f32 %7, %15, %8\n")
BODY(k_cvt, "v_cvt_f16_f32 %0, %8\n v_cvt_f16_f32 %1, %9\n v_cvt_f16_f32 %2, %10\n v_cvt_f16_f32 %3, %11\n"
            "v_cvt_f16_f32 %4, %12\n v_cvt_f16_f32 %5, %13\n v_cvt_f16_f32 %6, %14\n v_cvt_f16_f32 %7, %15\n")
BODY(k_pack, "v_pack_b32_f16 %0, %1, %2\n v_pack_b32_f16 %1, %2, %3\n v_pack_b32_f16 %2, %3, %4\n v_pack_b32_f16 %3, %4, %5\n"
             "v_pack_b32_f16 %4, %5, %6\n v_pack_b32_f16 %5, %6, %7\n v_pack_b32_f16 %6, %7, %0\n v_pack_b32_f16 %7, %0, %1\n")
BODY(k_pkmax, "v_pk_max_f16 %0, %0, 0\n v_pk_max_f16 %1, %1, 0\n v_pk_max_f16 %2, %2, 0\n v_pk_max_f16 %3, %3, 0\n"
              "v_pk_max_f16 %4, %4, 0\n v_pk_max_f16 %5, %5, 0\n v_pk_max_f16 %6, %6, 0\n v_pk_max_f16 %7, %7, 0\n")
BODY(k_and, "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n"
            "v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %0\n")
BODY(k_fmamix, "v_fma_mix_f32 %8, %0, 1.0, %8 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %9, %1, 1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %10, %2, 1.0, %10 op_sel_hi:[1,0,0]\n"
               "v_fma_mix_f32 %11, %3, 1.0, %11 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %12, %4, 1.0, %12 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %13, %5, 1.0, %13 op_sel_hi:[1,0,0]\n"
               "v_fma_mix_f32 %14, %6, 1.0, %14 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %15, %7, 1.0, %15 op_sel_hi:[1,0,0]\n")
BODY(k_fma, "v_fma_f32 %8, %8, %9, %10\n v_fma_f32 %9, %9, %10, %11\n v_fma_f32 %10, %10, %11, %12\n v_fma_f32 %11, %11, %12, %13\n"
            "v_fma_f32 %12, %12, %13, %14\n v_fma_f32 %13, %13, %14, %15\n v_fma_f32 %14, %14, %15, %8\n v_fma_f32 %15, %15, %8, %9\n")
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  const int iters = 20000;
#define RUN(k, n) { hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, 100); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, iters); hipDeviceSynchronize(); unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
    printf("%-22s %6.2f cycles per instruction (one wave per SIMD, 256 workgroups)\n", #k, (double)h[0] / iters / n); }
  RUN(k_cvt_pk, 8) RUN(k_cvt, 8) RUN(k_pack, 8) RUN(k_pkmax, 8) RUN(k_and, 8) RUN(k_fmamix, 8) RUN(k_fma, 8)
  return 0;
}
