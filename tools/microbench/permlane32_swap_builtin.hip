#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16x __attribute__((ext_vector_type(16)));
__global__ void k(unsigned* out, const float* in) {
  int lane = threadIdx.x;
  f16x acc;
  for (int i = 0; i < 16; ++i) acc[i] = in[lane * 16 + i];
  float v[8];
  for (int j = 0; j < 4; ++j) {
    const float fa = acc[0 * 4 + j], fb = acc[1 * 4 + j];
    unsigned qa = __builtin_bit_cast(unsigned, fa);
    unsigned qb = __builtin_bit_cast(unsigned, fb);
    const auto sw = __builtin_amdgcn_permlane32_swap(qa, qb, false, false);
    v[j] = __builtin_bit_cast(float, sw[0]);
    v[4 + j] = __builtin_bit_cast(float, sw[1]);
  }
  for (int j = 0; j < 8; ++j) out[lane * 8 + j] = (unsigned)v[j];
}
int main() {
  float h[64 * 16]; for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) h[l * 16 + i] = l * 100 + i;
  float* d; unsigned* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 8 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(o, d); unsigned r[64 * 8]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 32, 33}) { printf("lane %2d:", l); for (int j = 0; j < 8; ++j) printf(" %u", r[l * 8 + j]); printf("\n"); }
  return 0;
}
