// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/hbm_alias.hip -o tools/microbench/hbm_alias ; run it on the GPU box
// Micro-benchmark (round 4): does the read : write wall of the streaming mixes (tools/microbench/hbm_mix.hip: 2 : 1 = 4.6-5.7 TB/s)
// depend on how the streams' base addresses are aligned to each other?  The plan's workspace puts tensors of 120 MB ... 2.5 GB
// back to back; hbm_mix puts its arrays exactly 1 GiB apart.  Same kernel, the R read streams and the write stream staggered by
// `pad` bytes each (0 = the aliased layout).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int R>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, size_t stride)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 a = make_uint4(1, 2, 3, 4);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4 v = in[r * stride + i];
            a.x += v.x; a.y ^= v.y; a.z += v.z; a.w ^= v.w;
        }
        out[i] = a;
    }
}

template <int R>
static void run(const uint4* in, uint4* out, size_t n, size_t pad, hipEvent_t e0, hipEvent_t e1)
{
    const size_t stride = n + pad / 16;
    for (int g : {256, 512}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<R>), dim3(g), dim3(256), 0, 0, in, out + (R * pad) / 16, n, stride);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms / 4 < best) best = ms / 4;
        }
        const double bytes = (double)(R + 1) * n * 16;
        printf("read %d : write 1  stagger %8zu B  grid %4d  %.3f ms  %.2f TB/s\n", R, pad, g, best, bytes / (best * 1e-3) / 1e12);
    }
}

int main()
{
    const size_t n = (size_t)1 << 26;                        // 1 GiB per stream
    const size_t slack = (size_t)64 << 20;
    uint4 *in, *out;
    CK(hipMalloc(&in, 4 * n * 16 + slack));
    CK(hipMalloc(&out, n * 16 + slack));
    CK(hipMemset(in, 1, 4 * n * 16 + slack));
    CK(hipMemset(out, 0, n * 16 + slack));
    printf("in %p out %p\n", (void*)in, (void*)out);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t pad : {(size_t)0, (size_t)256, (size_t)4096 + 256, (size_t)65536 + 4096 + 256, ((size_t)1 << 20) + 65536 + 4096 + 256, ((size_t)5 << 20) + 768}) {
        run<2>(in, out, n, pad, e0, e1);
        run<4>(in, out, n, pad, e0, e1);
    }
    return 0;
}
