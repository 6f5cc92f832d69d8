// Visualisation / training-branch extras of FGAC (DeMFInet.py:454-496) and of DeMFInet.forward's extended return tuples (167-176): the
// per-pixel channel mean of |A| (or |A - B|), the global min-max normalisation of such a map, and 1 - w.  Debug imagery: plain
// point-wise kernels, deterministic (min / max are order-independent; the channel sum runs in channel order in fp32).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int VZ_PARTS = 256;                                    // partial (min, max) pairs of the normalisation

// out[y, x] = mean_c |a[y, x, c] - b[y, x, c]|  (b optional).  torch.mean(torch.abs(.), 1, keepdim=True) of 456-457, 465-466, 473-474,
// 481-482, 489-490: fp32 sum over the C channels divided by C.
template <typename T>
__global__ void absmean_kernel(demfi_view a, demfi_view b, float* __restrict__ out, int C, int H, int W)
{
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    const DEMFI_GLOBAL T* pa = gcp<T>(a.ptr) + y * a.sy + x * a.sx;
    float s = 0.0f;
    if (b.ptr) {
        const DEMFI_GLOBAL T* pb = gcp<T>(b.ptr) + y * b.sy + x * b.sx;
        for (int c = 0; c < C; ++c) s += fabsf((float)pa[c * a.sc] - (float)pb[c * b.sc]);
    } else {
        for (int c = 0; c < C; ++c) s += fabsf((float)pa[c * a.sc]);
    }
    out[i] = s / (float)C;
}

// pass 1: block k reduces its slice to (min, max)
__global__ void minmax_part_kernel(const float* __restrict__ p, int64_t n, float* __restrict__ parts)
{
    __shared__ float smin[NT], smax[NT];
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float v = p[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { parts[2 * blockIdx.x] = smin[0]; parts[2 * blockIdx.x + 1] = smax[0]; }
}

// pass 2: x -= min; x /= max(x - min)  -- the reference's two in-place steps (459-461): fl(fl(x - min) / fl(max - min))
__global__ void minmax_apply_kernel(float* __restrict__ p, int64_t n, const float* __restrict__ parts, int n_parts)
{
    __shared__ float smin[NT], smax[NT];
    float lo = INFINITY, hi = -INFINITY;
    for (int k = threadIdx.x; k < n_parts; k += NT) {
        lo = fminf(lo, parts[2 * k]);
        hi = fmaxf(hi, parts[2 * k + 1]);
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    const float mn = smin[0], den = smax[0] - smin[0];
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) p[i] = (p[i] - mn) / den;
}

__global__ void one_minus_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) out[i] = 1.0f - in[i];
}

}  // namespace

extern "C" int demfi_absmean_map(const demfi_view* a, const demfi_view* b, float* out, int C, int H, int W, void* stream)
{
    if (!a || !a->ptr || !out || C <= 0 || H <= 0 || W <= 0) return demfi_set_error(DEMFI_ERR_ARG, "demfi_absmean_map: bad args");
    if (b && b->ptr && b->is_f32 != a->is_f32) return demfi_set_error(DEMFI_ERR_ARG, "demfi_absmean_map: view type mismatch");
    demfi_view bv = {nullptr, 0, 0, 0, 0, 0, 0};
    if (b && b->ptr) bv = *b;
    const unsigned grid = (unsigned)(((int64_t)H * W + NT - 1) / NT);
    if (a->is_f32) hipLaunchKernelGGL(absmean_kernel<float>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, *a, bv, out, C, H, W);
    else           hipLaunchKernelGGL(absmean_kernel<half_t>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, *a, bv, out, C, H, W);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int64_t demfi_minmax_scratch_floats(void) { return 2 * VZ_PARTS; }

extern "C" int demfi_minmax_normalize(float* plane, int64_t n, float* scratch, void* stream)
{
    if (!plane || !scratch || n <= 0) return demfi_set_error(DEMFI_ERR_ARG, "demfi_minmax_normalize: bad args");
    const int64_t blocks = (n + NT - 1) / NT;
    const int parts = (int)(blocks < VZ_PARTS ? blocks : VZ_PARTS);
    hipLaunchKernelGGL(minmax_part_kernel, dim3(parts), dim3(NT), 0, (hipStream_t)stream, plane, n, scratch);
    hipLaunchKernelGGL(minmax_apply_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, plane, n, scratch, parts);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_one_minus(const float* in, float* out, int64_t n, void* stream)
{
    if (!in || !out || n <= 0) return demfi_set_error(DEMFI_ERR_ARG, "demfi_one_minus: bad args");
    hipLaunchKernelGGL(one_minus_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, in, out, n);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}
