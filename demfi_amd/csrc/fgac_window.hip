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

__device__ __forceinline__ float group8_sum(float v)
{
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__device__ __forceinline__ void bilerp8(const char* base, int64_t sx, int64_t sy, const SampleMap& m, int H, int W, float* o)
{
    uint4 raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yy = min(max(m.y0 + (k >> 1), 0), H - 1), xx = min(max(m.x0 + (k & 1), 0), W - 1);
        raw[k] = ld_global16(base + (int64_t)yy * sy + (int64_t)xx * sx);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const h8_t v = __builtin_bit_cast(h8_t, raw[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = o[j] + (float)v[j] * m.w[k];
    }
}

// fp16 NHWC views, C = 64.  R = 2 rr + 1 (template: the per-element values stay in registers).
template <int R, int MODE>
__global__ __launch_bounds__(FW_NT) void fgac_window_kernel(demfi_view REF, demfi_view SRC, const float* __restrict__ flow,
                                                            demfi_view O, int H, int W, float* __restrict__ attn)
{
    constexpr int E = R * R, rr = R / 2;
    // mode 1: per-pixel texel window [(R+1)*(R+1)][64 ch] fp16 = 128 B per texel, 32 pixels per workgroup (dynamic LDS:
    // 64 KiB for R = 3, 144 KiB for R = 5)
    extern __shared__ __attribute__((aligned(16))) char win[];
    const int64_t hw = (int64_t)H * W;
    const int64_t gi = (int64_t)blockIdx.x * FW_NT + threadIdx.x;
    const int64_t pix = gi >> 3;
    const int part = (int)(gi & 7);
    const bool live = pix < hw;
    const int y = live ? (int)(pix / W) : 0, x = live ? (int)(pix - (int64_t)y * W) : 0;
    const char* rbase = (const char*)REF.ptr + part * 16;
    const int64_t rsx = REF.sx * 2, rsy = REF.sy * 2;
    float sk[8];
    {
        const h8_t v = __builtin_bit_cast(h8_t, ld_global16((const char*)SRC.ptr + ((int64_t)y * SRC.sy + (int64_t)x * SRC.sx) * 2 + part * 16));
#pragma unroll
        for (int j = 0; j < 8; ++j) sk[j] = (float)v[j];
    }
    float g[E][8];
    float corr[E];
    if constexpr (MODE == 1) {
        // ---- stage the (R+1)^2 texel window around the centroid: one 16-byte slice per lane and texel -----------
        const float ix = unnormalized_coord(flow[y * (int64_t)W + x] - (float)rr, (float)(W - 1), (float)(W - 1));   // sample kj = 0
        const float iy = unnormalized_coord(flow[hw + y * (int64_t)W + x] - (float)rr, (float)(H - 1), (float)(H - 1));
        // the R samples along an axis are 1 texel apart only if the round trip is linear; it is not exactly, so
        // every sample's own coordinate is recomputed below and the window is addressed relative to the floor of sample 0
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int bx = (int)fminf(fmaxf(fx0, -8.0f), (float)W + 8.0f), by = (int)fminf(fmaxf(fy0, -8.0f), (float)H + 8.0f);
        char* mywin = win + (threadIdx.x >> 3) * ((R + 1) * (R + 1) * 128) + part * 16;
#pragma unroll
        for (int ty = 0; ty <= R; ++ty)
#pragma unroll
            for (int tx = 0; tx <= R; ++tx) {
                const int yy = by + ty, xx = bx + tx;
                uint4 v = make_uint4(0, 0, 0, 0);                            // zeros padding
                if (live && yy >= 0 && yy < H && xx >= 0 && xx < W) v = ld_global16(rbase + (int64_t)yy * rsy + (int64_t)xx * rsx);
                *(uint4*)(mywin + (ty * (R + 1) + tx) * 128) = v;
            }
        __builtin_amdgcn_wave_barrier();                                     // window rows are private to 8 lanes of one wave
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
                for (int j = 0; j < 8; ++j) g[e][j] = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!(m.inb & (1 << k))) continue;                       // out-of-image corner: weight 0
                    const int ty = m.y0 + (k >> 1) - by, tx = m.x0 + (k & 1) - bx;
                    h8_t v;
                    if (ty < 0 || ty > R || tx < 0 || tx > R)               // a floor() that moved by one through the round trip: rare, exact
                        v = __builtin_bit_cast(h8_t, ld_global16(rbase + (int64_t)(m.y0 + (k >> 1)) * rsy + (int64_t)(m.x0 + (k & 1)) * rsx));
                    else
                        v = *(const h8_t*)(mywin + (ty * (R + 1) + tx) * 128);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[e][j] = g[e][j] + (float)v[j] * m.w[k];
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
                for (int j = 0; j < 8; ++j) g[e][j] = 0.0f;
                if (r < 0 || r >= R * H || c < 0 || c >= R * W) continue;       // unfold's zero padding (427)
                const int i = r / H, h = r - i * H, jj = c / W, w = c - jj * W;
                const int fy = (h * R + i) % H, fx = (w * R + jj) % W;          // tiled centroid grid (411)
                const float px = flow[(int64_t)fy * W + fx] + (float)(i - rr);  // channel 0 (x) gets dy[i] (405-408)
                const float py = flow[hw + (int64_t)fy * W + fx] + (float)(jj - rr);
                const float sx_ = unnormalized_coord(px, (float)(W - 1), (float)(W - 1));
                const float sy_ = unnormalized_coord(py, (float)(H - 1), (float)(H - 1));
                const SampleMap m = make_sample_map(sx_, sy_, H, W);
                bilerp8(rbase, rsx, rsy, m, H, W, g[e]);
            }
    }
    // ---- correlation over the channels: 8 per lane, then across the 8 lanes of the pixel (wavefront shuffles) ----------
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s = s + g[e][j] * sk[j];
        corr[e] = group8_sum(s);
        mx = fmaxf(mx, corr[e]);
    }
    float den = 0.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) { corr[e] = expf(corr[e] - mx); den += corr[e]; }     // nn.Softmax(dim=1), 441
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const float a = corr[e] / den;
        if (attn && live && part == 0) attn[(int64_t)e * hw + pix] = a;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = o[j] + a * g[e][j];
    }
    if (live) {
        h8_t ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = (half_t)o[j];
        st_global16((char*)O.ptr + ((int64_t)y * O.sy + (int64_t)x * O.sx) * 2 + part * 16, __builtin_bit_cast(uint4, ov));
    }
}

// F.avg_pool2d(x, (P,P), (1,1), padding=sr) with count_include_pad (DeMFInet.py:417, 434): NHWC fp16, 16 bytes per lane.
__global__ void avg_pool_fat_kernel(demfi_view S, demfi_view O, int C, int H, int W, int sr)
{
    const int lpp = C / 8;
    const int64_t i = (int64_t)blockIdx.x * FW_NT + threadIdx.x;
    const int64_t pix = i / lpp;
    if (pix >= (int64_t)H * W) return;
    const int part = (int)(i - pix * lpp);
    const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    for (int dy = -sr; dy <= sr; ++dy)
        for (int dx = -sr; dx <= sr; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const h8_t v = __builtin_bit_cast(h8_t, ld_global16((const char*)S.ptr + ((int64_t)yy * S.sy + (int64_t)xx * S.sx) * 2 + part * 16));
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
    const float inv = 1.0f / (float)((2 * sr + 1) * (2 * sr + 1));
    h8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(acc[j] * inv);
    st_global16((char*)O.ptr + ((int64_t)y * O.sy + (int64_t)x * O.sx) * 2 + part * 16, __builtin_bit_cast(uint4, o));
}

bool fat16(const demfi_view* v) { return v && v->ptr && v->sc == 1 && !v->is_f32; }

}  // namespace

extern "C" int demfi_avg_pool_fat(const demfi_view* src, const demfi_view* out, int C, int H, int W, int sr, void* stream)
{
    if (!fat16(src) || !fat16(out) || C <= 0 || C % 8 || H <= 0 || W <= 0 || sr < 0 || sr > 4)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_avg_pool_fat: fp16 NHWC views, C %% 8 == 0, 0 <= sr <= 4");
    const int64_t n = (int64_t)H * W * (C / 8);
    hipLaunchKernelGGL(avg_pool_fat_kernel, dim3((unsigned)((n + FW_NT - 1) / FW_NT)), dim3(FW_NT), 0, (hipStream_t)stream, *src, *out, C, H, W, sr);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_fgac_window(const demfi_view* ref_k, const demfi_view* source_k, const float* flow, const demfi_view* out,
                                 int C, int H, int W, int rr, int mode, float* attn_out, void* stream)
{
    if (!fat16(ref_k) || !fat16(source_k) || !fat16(out) || !flow || C != 64 || H <= 1 || W <= 1)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_fgac_window: fp16 NHWC views with C = 64 expected");
    if (rr < 1 || rr > 2 || (mode != 0 && mode != 1))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_fgac_window: rr in {1, 2}, mode in {0, 1} (rr = 0 is demfi_fgac_gather)");
    if ((int64_t)H * W * (2 * rr + 1) >= (1ll << 30)) return demfi_set_error(DEMFI_ERR_ARG, "demfi_fgac_window: image too large");
    const int64_t n = (int64_t)H * W * 8;
    const dim3 grid((unsigned)((n + FW_NT - 1) / FW_NT)), blk(FW_NT);
    hipStream_t st = (hipStream_t)stream;
    const int R = 2 * rr + 1;
    const size_t lds = mode == 1 ? (size_t)32 * (R + 1) * (R + 1) * 128 : 0;
    DEMFI_LDS_ATTR((fgac_window_kernel<3, 1>));
    DEMFI_LDS_ATTR((fgac_window_kernel<5, 1>));
    if (rr == 1 && mode == 0) hipLaunchKernelGGL((fgac_window_kernel<3, 0>), grid, blk, lds, st, *ref_k, *source_k, flow, *out, H, W, attn_out);
    else if (rr == 1) hipLaunchKernelGGL((fgac_window_kernel<3, 1>), grid, blk, lds, st, *ref_k, *source_k, flow, *out, H, W, attn_out);
    else if (mode == 0) hipLaunchKernelGGL((fgac_window_kernel<5, 0>), grid, blk, lds, st, *ref_k, *source_k, flow, *out, H, W, attn_out);
    else hipLaunchKernelGGL((fgac_window_kernel<5, 1>), grid, blk, lds, st, *ref_k, *source_k, flow, *out, H, W, attn_out);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}
