// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/mall_bw.hip -o tools/microbench/mall_bw ; run it on the GPU box
// Micro-benchmark (round 4): what does a CHAIN of streaming launches reach when the tensors it hands from launch to launch are
// small enough to stay in the 256 MiB Infinity Cache?  Step j reads X[j % 4] (and X[(j + 3) % 4] as a "residual" when R = 2) and
// writes X[(j + 1) % 4]: the access pattern of a layer chain run over a BAND of rows (the band schedule of DESIGN section 5) with S
// bytes per band and tensor.  S = 1 GiB is the HBM reference (the pattern of today's batch-21 launches).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int R, bool NT>
__global__ __launch_bounds__(256) void k(const v4u* __restrict__ in0, const v4u* __restrict__ in1, v4u* __restrict__ out, size_t n)
{
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n; i += 4 * step) {
        v4u a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = NT ? __builtin_nontemporal_load(&in0[i + u * step]) : in0[i + u * step];
        if (R == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += NT ? __builtin_nontemporal_load(&in1[i + u * step]) : in1[i + u * step];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] += 1;
            if (NT) __builtin_nontemporal_store(a[u], &out[i + u * step]); else out[i + u * step] = a[u];
        }
    }
    for (; i < n; i += step) out[i] = in0[i] + 1;
}

template <int R, bool NT>
static void chain(v4u* base, size_t slot, size_t n, int grid, int steps, hipEvent_t e0, hipEvent_t e1, const char* tag)
{
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int j = 0; j < steps; ++j)
            hipLaunchKernelGGL((k<R, NT>), dim3(grid), dim3(256), 0, 0, base + (j % 4) * slot, base + ((j + 3) % 4) * slot, base + ((j + 1) % 4) * slot, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms / steps < best) best = ms / steps;
    }
    const double bytes = (double)(R + 1) * n * 16;
    printf("%-10s read %d : write 1  S %5zu MiB  grid %5d  %.4f ms/step  %.2f TB/s\n", tag, R, n * 16 >> 20, grid, best, bytes / (best * 1e-3) / 1e12);
}

// read-only re-reads of one S-byte buffer (does a READ allocate in the Infinity Cache?)
template <bool NT>
__global__ __launch_bounds__(256) void rd(const v4u* __restrict__ in, v4u* __restrict__ out, size_t n)
{
    const size_t step = (size_t)gridDim.x * 256;
    v4u acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n; i += 4 * step) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += NT ? __builtin_nontemporal_load(&in[i + u * step]) : in[i + u * step];
    }
    if (acc.x == 0x12345678u && acc.y == 77) out[threadIdx.x] = acc;
}

int main()
{
    const size_t slot = (size_t)1 << 26;                       // 1 GiB per slot, four slots
    v4u* base;
    CK(hipMalloc(&base, 4 * slot * 16));
    CK(hipMemset(base, 1, 4 * slot * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[] = {8, 16, 24, 32, 48, 64, 96, 128, 192, 1024};
    for (size_t mb : sizes) {
        const size_t n = mb << 16;                             // MiB -> 16-byte elements
        const int steps = mb >= 512 ? 8 : 48;
        for (int g : {512, 1024, 2048}) {
            chain<1, false>(base, slot, n, g, steps, e0, e1, "plain");
            chain<2, false>(base, slot, n, g, steps, e0, e1, "plain");
        }
        chain<1, true>(base, slot, n, 512, steps, e0, e1, "nt");
        chain<2, true>(base, slot, n, 512, steps, e0, e1, "nt");
        for (int g : {512, 2048}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                for (int j = 0; j < steps; ++j) hipLaunchKernelGGL((rd<false>), dim3(g), dim3(256), 0, 0, base, base + 3 * slot, n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms / steps < best) best = ms / steps;
            }
            printf("re-read    S %5zu MiB  grid %5d  %.4f ms/step  %.2f TB/s\n", mb, g, best, (double)n * 16 / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
