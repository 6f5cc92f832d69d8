// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/hbm_mix.hip -o tools/microbench/hbm_mix ; run it on the GPU box
// Micro-benchmark: what does HBM deliver for a STREAMING kernel that reads R and writes W distinct 1-GiB arrays (16 B per lane,
// whole 128-byte lines per 8 lanes, grid-stride, 2 048 workgroups of 256 threads)?  The memory-bound kernels of this repository
// are all read + write mixes (SepConvGRU q layer 4 : 1, the residual 64 -> 64 convolution 2.3 : 1, warp_blend ~3 : 1): the
// ceiling to price them against is the mix's, not the 8 TB/s pin rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int R, int W>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, size_t stride)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 a = make_uint4(1, 2, 3, 4);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4 v = in[r * stride + i];
            a.x += v.x; a.y ^= v.y; a.z += v.z; a.w ^= v.w;
        }
#pragma unroll
        for (int w = 0; w < W; ++w) out[w * stride + i] = a;
        if (W == 0 && a.x == 0x12345678u && a.y == 77) out[i] = a;      // keeps the loads alive
    }
}

// variant: non-temporal stores and loads, 4 lines of every stream in flight per thread
template <int R, int W>
__global__ __launch_bounds__(256) void knt(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, size_t stride)
{
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * step < n; i += 4 * step) {
        uint4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = make_uint4(1, 2, 3, 4);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const v4u v = __builtin_nontemporal_load((const v4u*)&in[r * stride + i + u * step]);
                a[u].x += v.x; a[u].y ^= v.y; a[u].z += v.z; a[u].w ^= v.w;
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
#pragma unroll
            for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v4u{a[u].x, a[u].y, a[u].z, a[u].w}, (v4u*)&out[w * stride + i + u * step]);
        }
        if (W == 0 && a[0].x == 0x12345678u && a[1].y == 77 && a[2].z == 5 && a[3].w == 9) out[i] = a[0];
    }
}

template <int R, int W>
static void run_nt(const uint4* in, uint4* out, size_t n, size_t stride, hipEvent_t e0, hipEvent_t e1)
{
    for (int g : {256, 512, 1024}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((knt<R, W>), dim3(g), dim3(256), 0, 0, in, out, n, stride);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms / 4 < best) best = ms / 4;
        }
        const double bytes = (double)(R + W) * n * 16;
        printf("nt, 4 deep: read %d : write %d  grid %5d  %.3f ms  %.2f TB/s  (%.3f of 8 TB/s)\n", R, W, g, best, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 8e12);
    }
}

template <int R, int W>
static void run(const uint4* in, uint4* out, size_t n, size_t stride, hipEvent_t e0, hipEvent_t e1)
{
    for (int g : {256, 512, 2048}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<R, W>), dim3(g), dim3(256), 0, 0, in, out, n, stride);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms / 4 < best) best = ms / 4;
        }
        const double bytes = (double)(R + W) * n * 16;
        printf("read %d : write %d  grid %5d  %.3f ms  %.2f TB/s  (%.3f of 8 TB/s)\n", R, W, g, best, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 8e12);
    }
}

int main()
{
    const size_t n = (size_t)1 << 26;                        // 1 GiB per array
    uint4 *in, *out;
    CK(hipMalloc(&in, 4 * n * 16));
    CK(hipMalloc(&out, 2 * n * 16));
    CK(hipMemset(in, 1, 4 * n * 16));
    CK(hipMemset(out, 0, 2 * n * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    run<1, 0>(in, out, n, n, e0, e1);
    run<4, 0>(in, out, n, n, e0, e1);
    run<0, 1>(in, out, n, n, e0, e1);
    run<1, 1>(in, out, n, n, e0, e1);
    run<2, 1>(in, out, n, n, e0, e1);
    run<3, 1>(in, out, n, n, e0, e1);
    run<4, 1>(in, out, n, n, e0, e1);
    run_nt<1, 0>(in, out, n, n, e0, e1);
    run_nt<0, 1>(in, out, n, n, e0, e1);
    run_nt<1, 1>(in, out, n, n, e0, e1);
    run_nt<3, 1>(in, out, n, n, e0, e1);
    run_nt<4, 1>(in, out, n, n, e0, e1);
    return 0;
}
