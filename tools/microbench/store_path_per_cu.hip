// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/store_path_per_cu.hip -o tools/microbench/store_path_per_cu ; run it on the GPU box
// Micro-benchmark: how fast can 256 persistent workgroups (4 waves each) write an NHWC fp16 64-channel tensor (128-byte pixel
// records, 3 x 736 x 1280) in 8x32-pixel tiles, depending on how the 16-byte pieces of a store instruction are laid out?
//   0: (lx, hi) -> pixel lx, 16 B at q*32 + hi*16            (the persistent conv kernel's epilogue: 32-byte runs per pixel)
//   1: lane l   -> pixel l>>3, 16 B at (l&7)*16               (8 full 128-byte lines per instruction)
//   2: (lx, hi) -> pixel lx, 16 B at hi*64 + q*16             (16-byte pieces, 4 instructions fill a line half)
//   3: dword stores: (lx, hi) -> pixel 2*j+hi, 4 B at lx*4   (2 full lines per instruction, 16 x more instructions)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(char* out, int H, int W, int B)
{
    const int tx = (W + 31) / 32, ty = (H + 7) / 8, total = tx * ty * B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = lane & 31, hi = lane >> 5;
    uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int b = t / (tx * ty), r = t % (tx * ty), oy0 = (r / tx) * 8, ox0 = (r % tx) * 32;
        for (int p = 0; p < 2; ++p) {
            const int oy = oy0 + wave * 2 + p;
            char* row = out + (((size_t)b * H + oy) * W + ox0) * 128;
            if (MODE == 0) {
                for (int q = 0; q < 4; ++q) *(uint4*)(row + lx * 128 + q * 32 + hi * 16) = v;
            } else if (MODE == 1) {
                for (int q = 0; q < 4; ++q) *(uint4*)(row + (q * 8 + (lane >> 3)) * 128 + (lane & 7) * 16) = v;
            } else if (MODE == 2) {
                for (int q = 0; q < 4; ++q) *(uint4*)(row + lx * 128 + hi * 64 + q * 16) = v;
            } else {
                for (int j = 0; j < 16; ++j) *(unsigned*)(row + (2 * j + hi) * 128 + lx * 4) = v.x;
            }
        }
        __syncthreads();
    }
}

int main()
{
    const int H = 736, W = 1280, B = 3;
    const size_t bytes = (size_t)B * H * W * 128;
    char* out;
    CK(hipMalloc(&out, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int g = 256; g >= 16; g /= 2) {          // fewer workgroups: is ~10 B/clk a per-CU or a chip-wide limit?
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<0>, dim3(g), dim3(256), 0, 0, out, H, W, B);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("mode 0, %3d workgroups: %.4f ms, %.2f TB/s, %.1f B/clk/CU at 2.4 GHz\n", g, ms / 20, bytes / (ms / 20 * 1e-3) / 1e12, bytes / (ms / 20 * 1e-3) / g / 2.4e9);
        }
    }
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, H, W, B);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, H, W, B);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, H, W, B);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, H, W, B);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("mode %d: %.4f ms per launch, %.2f TB/s\n", mode, ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
        }
    }
    return 0;
}
