#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
// pattern 0: full lines (8 lanes x 16 B per 128-B pixel record), 1: half lines (4 lanes x 16 B, first halves then second halves)
__global__ void k(u4* out, long npx, int pattern, int reps) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long nthreads = (long)gridDim.x * blockDim.x;
  u4 v = {1u, 2u, 3u, (unsigned)tid};
  for (int r = 0; r < reps; ++r) {
    if (pattern == 0) {
      for (long i = tid; i < npx * 8; i += nthreads) out[i] = v;
    } else if (pattern == 1) {
      for (int half = 0; half < 2; ++half)
        for (long i = tid; i < npx * 4; i += nthreads) { long px = i >> 2; int q = i & 3; out[px * 8 + half * 4 + q] = v; }
    } else if (pattern == 3) {   // MFMA-layout stores after permlane swap: 32 px per wave, 32-byte runs, 4 instructions per line
      long wave = tid >> 6; int lane = tid & 63; long nwaves = nthreads >> 6;
      for (long blk = wave; blk < npx / 32; blk += nwaves)
        for (int k = 0; k < 4; ++k) { long px = blk * 32 + (lane & 31); out[px * 8 + k * 2 + (lane >> 5)] = v; }
    } else {   // pattern 2: like the conv epilogue: per wave 64 consecutive px, pass = 16 px, sub 0 then sub 1
      long wave = tid >> 6; int lane = tid & 63; long nwaves = nthreads >> 6;
      for (long blk = wave; blk < npx / 64; blk += nwaves)
        for (int s = 0; s < 2; ++s)
          for (int pass = 0; pass < 4; ++pass) { long px = blk * 64 + pass * 16 + (lane >> 2); out[px * 8 + s * 4 + (lane & 3)] = v; }
    }
  }
}
int main() {
  long npx = 736L * 1280 * 3; u4* d; hipMalloc(&d, npx * 128);
  for (int grid : {1024, 4096, 16384})
  for (int p = 0; p < 4; ++p) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<grid, 256>>>(d, npx, p, 1); hipDeviceSynchronize();
    hipEventRecord(a); k<<<grid, 256>>>(d, npx, p, 5); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("grid %5d pattern %d: %.3f ms per pass, %.1f GB/s\n", grid, p, ms / 5, npx * 128.0 / (ms / 5) / 1e6);
  }
  return 0;
}
