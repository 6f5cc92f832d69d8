// Generalised flow-guided attentive correlation (FGAC.forward with radii rr, sr > 0, /root/reference/DeMFInet.py:401-445;
// SURVEY.md section 8f rank 4).  The released code fixes rr = sr = 0 (point-wise FGAC = demfi_fgac_gather); this file is the
// window form: per output pixel a correlation volume over (2rr+1)^2 bilinear samples of ref_k, a softmax over the window
// and the attention-weighted sum (Eq. 3).
//
//   corr[e]  = sum_c G[c,e] * source_k[c]                (DeMFInet.py:438)
//   a[e]     = softmax_e(corr)                            (441)
//   out[c]   = sum_e a[e] * G[c,e]                        (443)
//
// G[c,e] = bilinear sample (zeros padding, align_corners=True, coordinates normalised and un-normalised like
// bilinear_sampler 499-508) of ref_k at an ABSOLUTE position flow[src(e)] + offset(e).  Two index maps:
//   mode 0 (REFERENCE): the map the reference code really computes when its radii are overridden -- the centroid grid is
//       tiled (`repeat`, 411) while the offsets are interleaved (`view/repeat`, 407-408) and the window is re-gathered by a
//       strided unfold (423-429), so element e = (ki,kj) of pixel (y,x) reads
//           r = y*R - rr + ki,  c = x*R - rr + kj      (zero, but still in the softmax, when outside [0,R*H) x [0,R*W))
//           (i,h) = divmod(r, H), (j,w) = divmod(c, W)
//           position = flow[(h*R + i) % H, (w*R + j) % W] + (i - rr, j - rr)
//       -- pinned by fixtures generated from a patched in-memory copy of the reference function (tools/make_goldens.py);
//   mode 1 (LOCAL): the window the paper describes -- position = flow[y,x] + (kj - rr, ki - rr).  All R^2 samples of a
//       pixel share their fractional parts, so the (R+1)x(R+1) texel window around the centroid is staged ONCE per pixel in
//       LDS and every sample is blended from it (4 R^2 gathers -> (R+1)^2); parity: oracle only (the reference never runs it).
// Mapping: 8 lanes per pixel (16 bytes = 8 fp16 channels each, C = 64), 8 pixels per wave; the channel reduction of corr and
// the softmax use wavefront shuffles (DPP row_shr / ds_swizzle through __shfl_xor) -- no LDS, no atomics.
#include "common.h"

namespace {

constexpr int FW_NT = 256;
constexpr int FW_MAXR = 5;                       // rr <= 2

// sum over the LPP lanes of a pixel (8 for fp16, 16 for fp32: 16 bytes of channels per lane)
template <int LPP>
__device__ __forceinline__ float group_sum(float v)
{
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if constexpr (LPP == 16) v += __shfl_xor(v, 8, 64);
    return v;
}

// the N = 16 / sizeof(T) channels of a 16-byte slice as floats, and back
template <typename T> struct Slice {
    static constexpr int N = 16 / sizeof(T);
    static __device__ __forceinline__ void unpack(const uint4& raw, float* o)
    {
        if constexpr (sizeof(T) == 2) {
            const h8_t v = __builtin_bit_cast(h8_t, raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
        } else {
            const f4_t v = __builtin_bit_cast(f4_t, raw);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = v[j];
        }
    }
    static __device__ __forceinline__ uint4 pack(const float* o)
    {
        if constexpr (sizeof(T) == 2) {
            h8_t v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (half_t)o[j];
            return __builtin_bit_cast(uint4, v);
        } else {
            f4_t v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = o[j];
            return __builtin_bit_cast(uint4, v);
        }
    }
};

template <typename T>
__device__ __forceinline__ void bilerp_slice(const char* base, int64_t sx, int64_t sy, const SampleMap& m, int H, int W, float* o)
{
    constexpr int N = Slice<T>::N;
    uint4 raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yy = min(max(m.y0 + (k >> 1), 0), H - 1), xx = min(max(m.x0 + (k & 1), 0), W - 1);
        raw[k] = ld_global16(base + (int64_t)yy * sy + (int64_t)xx * sx);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v[N];
        Slice<T>::unpack(raw[k], v);
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = o[j] + v[j] * m.w[k];
    }
}

// NHWC views of the path dtype T (fp16, or fp32 for the strict-parity configuration), C = 64.  R = 2 rr + 1 (template: the
// per-element values stay in registers).  LPP = 64 * sizeof(T) / 16 lanes per pixel, N = 16 / sizeof(T) channels per lane.
template <typename T, int R, int MODE>
__global__ __launch_bounds__(FW_NT) void fgac_window_kernel(demfi_view REF, demfi_view SRC, const float* __restrict__ flow,
                                                            demfi_view O, int H, int W, float* __restrict__ attn)
{
    constexpr int E = R * R, rr = R / 2;
    constexpr int N = Slice<T>::N, ESZ = sizeof(T), LPP = 64 * ESZ / 16, LSH = ESZ == 2 ? 3 : 4, TEX = 64 * ESZ;
    // mode 1: per-pixel texel window [(R+1)*(R+1)][64 ch] = TEX bytes per texel, FW_NT / LPP pixels per workgroup (dynamic LDS:
    // fp16 64 KiB for R = 3, 144 KiB for R = 5; fp32: the same -- half the pixels, twice the texel)
    extern __shared__ __attribute__((aligned(16))) char win[];
    const int64_t hw = (int64_t)H * W;
    const int64_t gi = (int64_t)blockIdx.x * FW_NT + threadIdx.x;
    const int64_t pix = gi >> LSH;
    const int part = (int)(gi & (LPP - 1));
    const bool live = pix < hw;
    const int y = live ? (int)(pix / W) : 0, x = live ? (int)(pix - (int64_t)y * W) : 0;
    const char* rbase = (const char*)REF.ptr + part * 16;
    const int64_t rsx = REF.sx * ESZ, rsy = REF.sy * ESZ;
    float sk[N];
    Slice<T>::unpack(ld_global16((const char*)SRC.ptr + ((int64_t)y * SRC.sy + (int64_t)x * SRC.sx) * ESZ + part * 16), sk);
    float g[E][N];
    float corr[E];
    if constexpr (MODE == 1) {
        // ---- stage the (R+1)^2 texel window around the centroid: one 16-byte slice per lane and texel -----------
        const float ix = unnormalized_coord(flow[y * (int64_t)W + x] - (float)rr, (float)(W - 1), (float)(W - 1));   // sample kj = 0
        const float iy = unnormalized_coord(flow[hw + y * (int64_t)W + x] - (float)rr, (float)(H - 1), (float)(H - 1));
        // the R samples along an axis are 1 texel apart only if the round trip is linear; it is not exactly, so
        // every sample's own coordinate is recomputed below and the window is addressed relative to the floor of sample 0
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int bx = (int)fminf(fmaxf(fx0, -8.0f), (float)W + 8.0f), by = (int)fminf(fmaxf(fy0, -8.0f), (float)H + 8.0f);
        char* mywin = win + (threadIdx.x >> LSH) * ((R + 1) * (R + 1) * TEX) + part * 16;
#pragma unroll
        for (int ty = 0; ty <= R; ++ty)
#pragma unroll
            for (int tx = 0; tx <= R; ++tx) {
                const int yy = by + ty, xx = bx + tx;
                uint4 v = make_uint4(0, 0, 0, 0);                            // zeros padding
                if (live && yy >= 0 && yy < H && xx >= 0 && xx < W) v = ld_global16(rbase + (int64_t)yy * rsy + (int64_t)xx * rsx);
                *(uint4*)(mywin + (ty * (R + 1) + tx) * TEX) = v;
            }
        __builtin_amdgcn_wave_barrier();                                     // window rows are private to the LPP lanes of a pixel (one wave)
#pragma unroll
        for (int ki = 0; ki < R; ++ki)
#pragma unroll
            for (int kj = 0; kj < R; ++kj) {
                const int e = ki * R + kj;
                const float sx_ = unnormalized_coord(flow[y * (int64_t)W + x] + (float)(kj - rr), (float)(W - 1), (float)(W - 1));
                const float sy_ = unnormalized_coord(flow[hw + y * (int64_t)W + x] + (float)(ki - rr), (float)(H - 1), (float)(H - 1));
                const SampleMap m = make_sample_map(sx_, sy_, H, W);
                // texel coordinates relative to the staged window; a sample whose floor falls outside it (possible only through
                // the non-linear round trip at huge coordinates) contributes its in-window corners, the rest is zero = padding
#pragma unroll
                for (int j = 0; j < N; ++j) g[e][j] = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!(m.inb & (1 << k))) continue;                       // out-of-image corner: weight 0
                    const int ty = m.y0 + (k >> 1) - by, tx = m.x0 + (k & 1) - bx;
                    uint4 raw;
                    if (ty < 0 || ty > R || tx < 0 || tx > R)               // a floor() that moved by one through the round trip: rare, exact
                        raw = ld_global16(rbase + (int64_t)(m.y0 + (k >> 1)) * rsy + (int64_t)(m.x0 + (k & 1)) * rsx);
                    else
                        raw = *(const uint4*)(mywin + (ty * (R + 1) + tx) * TEX);
                    float v[N];
                    Slice<T>::unpack(raw, v);
#pragma unroll
                    for (int j = 0; j < N; ++j) g[e][j] = g[e][j] + v[j] * m.w[k];
                }
            }
    } else {
#pragma unroll
        for (int ki = 0; ki < R; ++ki)
#pragma unroll
            for (int kj = 0; kj < R; ++kj) {
                const int e = ki * R + kj;
                const int r = y * R - rr + ki, c = x * R - rr + kj;
#pragma unroll
                for (int j = 0; j < N; ++j) g[e][j] = 0.0f;
                if (r < 0 || r >= R * H || c < 0 || c >= R * W) continue;       // unfold's zero padding (427)
                const int i = r / H, h = r - i * H, jj = c / W, w = c - jj * W;
                const int fy = (h * R + i) % H, fx = (w * R + jj) % W;          // tiled centroid grid (411)
                const float px = flow[(int64_t)fy * W + fx] + (float)(i - rr);  // channel 0 (x) gets dy[i] (405-408)
                const float py = flow[hw + (int64_t)fy * W + fx] + (float)(jj - rr);
                const float sx_ = unnormalized_coord(px, (float)(W - 1), (float)(W - 1));
                const float sy_ = unnormalized_coord(py, (float)(H - 1), (float)(H - 1));
                const SampleMap m = make_sample_map(sx_, sy_, H, W);
                bilerp_slice<T>(rbase, rsx, rsy, m, H, W, g[e]);
            }
    }
    // ---- correlation over the channels: 8 per lane, then across the 8 lanes of the pixel (wavefront shuffles) ----------
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < N; ++j) s = s + g[e][j] * sk[j];
        corr[e] = group_sum<LPP>(s);
        mx = fmaxf(mx, corr[e]);
    }
    float den = 0.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) { corr[e] = expf(corr[e] - mx); den += corr[e]; }     // nn.Softmax(dim=1), 441
    float o[N];
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = 0.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const float a = corr[e] / den;
        if (attn && live && part == 0) attn[(int64_t)e * hw + pix] = a;
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = o[j] + a * g[e][j];
    }
    if (live) st_global16((char*)O.ptr + ((int64_t)y * O.sy + (int64_t)x * O.sx) * ESZ + part * 16, Slice<T>::pack(o));
}

// F.avg_pool2d(x, (P,P), (1,1), padding=sr) with count_include_pad (DeMFInet.py:417, 434): NHWC of the path dtype, 16 bytes per lane.
template <typename T>
__global__ void avg_pool_fat_kernel(demfi_view S, demfi_view O, int C, int H, int W, int sr)
{
    constexpr int N = Slice<T>::N, ESZ = sizeof(T);
    const int lpp = C / N;
    const int64_t i = (int64_t)blockIdx.x * FW_NT + threadIdx.x;
    const int64_t pix = i / lpp;
    if (pix >= (int64_t)H * W) return;
    const int part = (int)(i - pix * lpp);
    const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.0f;
    for (int dy = -sr; dy <= sr; ++dy)
        for (int dx = -sr; dx <= sr; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            float v[N];
            Slice<T>::unpack(ld_global16((const char*)S.ptr + ((int64_t)yy * S.sy + (int64_t)xx * S.sx) * ESZ + part * 16), v);
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] += v[j];
        }
    // ATen: sum / pool_size with count_include_pad (a division, not a multiplication by the reciprocal: exact for fp32 parity)
    const float cnt = (float)((2 * sr + 1) * (2 * sr + 1));
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = acc[j] / cnt;
    st_global16((char*)O.ptr + ((int64_t)y * O.sy + (int64_t)x * O.sx) * ESZ + part * 16, Slice<T>::pack(acc));
}

bool fat_view(const demfi_view* v) { return v && v->ptr && v->sc == 1; }

template <typename T>
int launch_window(const demfi_view* ref_k, const demfi_view* source_k, const float* flow, const demfi_view* out, int H, int W, int rr, int mode,
                  float* attn_out, hipStream_t st)
{
    constexpr int LPP = 64 * sizeof(T) / 16;
    const int64_t n = (int64_t)H * W * LPP;
    const dim3 grid((unsigned)((n + FW_NT - 1) / FW_NT)), blk(FW_NT);
    const int R = 2 * rr + 1;
    const size_t lds = mode == 1 ? (size_t)(FW_NT / LPP) * (R + 1) * (R + 1) * 64 * sizeof(T) : 0;
    DEMFI_LDS_ATTR((fgac_window_kernel<T, 3, 1>));
    DEMFI_LDS_ATTR((fgac_window_kernel<T, 5, 1>));
    if (rr == 1 && mode == 0) hipLaunchKernelGGL((fgac_window_kernel<T, 3, 0>), grid, blk, lds, st, *ref_k, *source_k, flow, *out, H, W, attn_out);
    else if (rr == 1) hipLaunchKernelGGL((fgac_window_kernel<T, 3, 1>), grid, blk, lds, st, *ref_k, *source_k, flow, *out, H, W, attn_out);
    else if (mode == 0) hipLaunchKernelGGL((fgac_window_kernel<T, 5, 0>), grid, blk, lds, st, *ref_k, *source_k, flow, *out, H, W, attn_out);
    else hipLaunchKernelGGL((fgac_window_kernel<T, 5, 1>), grid, blk, lds, st, *ref_k, *source_k, flow, *out, H, W, attn_out);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

}  // namespace

extern "C" int demfi_avg_pool_fat(const demfi_view* src, const demfi_view* out, int C, int H, int W, int sr, void* stream)
{
    if (!fat_view(src) || !fat_view(out) || src->is_f32 != out->is_f32 || C <= 0 || C % 8 || H <= 0 || W <= 0 || sr < 0 || sr > 4)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_avg_pool_fat: NHWC views of one dtype, C %% 8 == 0, 0 <= sr <= 4");
    const int per = src->is_f32 ? 4 : 8;
    const int64_t n = (int64_t)H * W * (C / per);
    const dim3 grid((unsigned)((n + FW_NT - 1) / FW_NT));
    if (src->is_f32) hipLaunchKernelGGL(avg_pool_fat_kernel<float>, grid, dim3(FW_NT), 0, (hipStream_t)stream, *src, *out, C, H, W, sr);
    else hipLaunchKernelGGL(avg_pool_fat_kernel<half_t>, grid, dim3(FW_NT), 0, (hipStream_t)stream, *src, *out, C, H, W, sr);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_fgac_window(const demfi_view* ref_k, const demfi_view* source_k, const float* flow, const demfi_view* out,
                                 int C, int H, int W, int rr, int mode, float* attn_out, void* stream)
{
    if (!fat_view(ref_k) || !fat_view(source_k) || !fat_view(out) || !flow || C != 64 || H <= 1 || W <= 1 ||
        ref_k->is_f32 != source_k->is_f32 || ref_k->is_f32 != out->is_f32)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_fgac_window: NHWC views of one dtype (fp16 or fp32) with C = 64 expected");
    if (rr < 1 || rr > 2 || (mode != 0 && mode != 1))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_fgac_window: rr in {1, 2}, mode in {0, 1} (rr = 0 is demfi_fgac_gather)");
    if ((int64_t)H * W * (2 * rr + 1) >= (1ll << 30)) return demfi_set_error(DEMFI_ERR_ARG, "demfi_fgac_window: image too large");
    if (ref_k->is_f32) return launch_window<float>(ref_k, source_k, flow, out, H, W, rr, mode, attn_out, (hipStream_t)stream);
    return launch_window<half_t>(ref_k, source_k, flow, out, H, W, rr, mode, attn_out, (hipStream_t)stream);
}
