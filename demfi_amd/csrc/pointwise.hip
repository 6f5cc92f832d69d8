// Memory-bound kernels of the DeMFI-Net_rb forward path for gfx950: space-to-depth, reflect pad, the
// complementary-flow-reversal splat, backward warp + occlusion blend, FGAC gather and the Eq.(4) gate blend.
// Built with -ffp-contract=off: the coordinate arithmetic must round exactly like the reference's
// step-by-step fp32 tensor ops so that the integer index / validity maps are bit-identical.
#include "common.h"
#include <stdlib.h>

namespace {

// pointer of per-t context q in a batched launch: base + q * (byte stride); NULL stays NULL
template <typename P> __device__ __forceinline__ P* bofs(P* p, int64_t bytes) { return p ? (P*)((const char*)p + bytes) : p; }

constexpr int NT = 256;

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + NT - 1) / NT); }

// ------------------------------------------------------------------------------------------------------
// pixel_reshuffle(cat(B0,B1,B-1,B2), 2)  (DeMFInet.py:234-235, 290-316)
// thread = (plane fc = frame*3 + c, h, w): reads the 2x2 block of one plane, writes 4 adjacent channels.
template <typename T>
__global__ void s2d_kernel(const float* __restrict__ x, T* __restrict__ out, int H, int W)
{
    const int H2 = H >> 1, W2 = W >> 1;
    const int64_t n = (int64_t)12 * H2 * W2;
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const int w = (int)(i % W2);
    const int h = (int)((i / W2) % H2);
    const int fc = (int)(i / ((int64_t)W2 * H2));
    const int f = fc / 3, c = fc - f * 3;
    const float* p = x + ((int64_t)(c * 4 + f) * H + 2 * h) * W + 2 * w;     // x is [3,4,H,W]
    const float2 r0 = *(const float2*)p;
    const float2 r1 = *(const float2*)(p + W);
    T* o = out + ((int64_t)h * W2 + w) * 48 + fc * 4;
    o[0] = (T)r0.x; o[1] = (T)r0.y; o[2] = (T)r1.x; o[3] = (T)r1.y;
}

// F.pad(mode="reflect") bottom/right (utils.py:1363)
__global__ void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ out, int planes, int h, int w,
                                   int H, int W)
{
    const int64_t n = (int64_t)planes * H * W;
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const int X = (int)(i % W);
    const int Y = (int)((i / W) % H);
    const int p = (int)(i / ((int64_t)W * H));
    const int sx = X < w ? X : 2 * (w - 1) - X;
    const int sy = Y < h ? Y : 2 * (h - 1) - Y;
    out[i] = x[((int64_t)p * h + sy) * w + sx];
}

// torch.mean(x[:, :, 0:2], dim=2)  (DeMFInet.py:178)
__global__ void overlay_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t hw)
{
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= 3 * hw) return;
    const int c = (int)(i / hw);
    const int64_t r = i - c * hw;
    out[i] = (x[(int64_t)(c * 4) * hw + r] + x[(int64_t)(c * 4 + 1) * hw + r]) / 2.0f;
}

// ------------------------------------------------------------------------------------------------------
// Forward splat of CFR (fwarp / sample_one, DeMFInet.py:625-729) + the linear combination / normalisation
// (CFR_flow_t_align, 606-622).
//
// Sums are accumulated in 2^-32 fixed point (64-bit integers): exact, hence independent of the order of the adds and
// bit-reproducible run to run (the reference's GPU path uses float atomics).  Round 1 did all 24 adds per source pixel
// as device-scope atomics -- those bypass the XCD L2s and run at fabric rate (0.59 ms at 720p, 0.6 % of the HBM
// roofline).  Now the work is partitioned by TARGET: one workgroup owns a CFR_TH x CFR_TW tile of target pixels, scans
// the source pixels of the tile + a halo of CFR_R pixels, and accumulates the contributions that land in its tile with
// LDS atomics (ds_add_u64); the finish (Eq. 614-620) runs from LDS and writes each output once.  No global accumulator
// is touched on this path.  Exactness for arbitrary flows: a source whose scaled displacement has floor() outside
// [-CFR_R, CFR_R-1] ("far", rare) cannot be seen by every tile it hits; cfr_far_kernel (one thread per source, which
// also writes the debug index maps) adds exactly those sources into the global int64 planes and flags the target
// tiles; a flagged tile adds its slice of the planes in the finish and leaves it zeroed.  near/far is a function of
// the source value alone, so every source is counted exactly once.
constexpr double FIX = 4294967296.0;
constexpr int CFR_TH = 16, CFR_TW = 64, CFR_R = 32;

struct CfrSrc {
    float v0, v1, fx, fy, x1, y1;
    bool far_;
};

__device__ __forceinline__ CfrSrc cfr_source(const float* __restrict__ fl, int64_t hw, int64_t pix, float sc)
{
    CfrSrc s;
    s.v0 = fl[pix];
    s.v1 = fl[hw + pix];
    s.fy = sc * s.v0;                                        // "y": column displacement (flo[:,0], 647)
    s.fx = sc * s.v1;                                        // "x": row displacement    (flo[:,1], 648)
    s.x1 = floorf(s.fx);
    s.y1 = floorf(s.fy);
    // !(a >= lo && a <= hi) is also true for NaN: such sources take the far path, whose range test drops them
    s.far_ = !(s.x1 >= (float)-CFR_R && s.x1 <= (float)(CFR_R - 1) && s.y1 >= (float)-CFR_R && s.y1 <= (float)(CFR_R - 1));
    return s;
}

// flat target index of corner cidx ((x1,y1) (x1,y2) (x2,y1) (x2,y2): 663-666) of a source at (y, x), -1 when masked (716)
__device__ __forceinline__ int cfr_target(const CfrSrc& s, int cidx, int y, int x, int H, int W, float& w)
{
    const float xs = s.x1 + (float)(cidx >> 1), ys = s.y1 + (float)(cidx & 1);
    const float dx = s.fx - xs, dy = s.fy - ys;
    w = expf(-(dx * dx + dy * dy));                          // get_gaussian_weights, 674-680
    if (fabsf(xs) < 1.0e9f && fabsf(ys) < 1.0e9f) {
        const long long r = (long long)xs + y, c = (long long)ys + x;     // idxx / idxy, 712-713
        if (r >= 0 && r < H && c >= 0 && c < W) return (int)(r * W + c);  // mask, 716
    }
    return -1;
}

__device__ __forceinline__ unsigned long long cfr_fix(float p) { return (unsigned long long)__double2ll_rn((double)p * FIX); }

// One thread per SOURCE pixel and flow: debug index maps for every source, global accumulation for the far ones only.
struct CfrBatch { int64_t f01, f10, t, acc, out, logit, pack; };   // byte strides between per-t contexts (blockIdx.y)
// optional packed copy written by the finish (round 6): the NHWC record Refine_Module.enc1 stages with vector loads -- 16 channels of
// the path dtype = [flow_t0 (2), flow_t1 (2) | flow_01 (2), flow_10 (2), occlusion logit | 7 zeros] (the thin members of Agg1,
// DeMFInet.py:77), i.e. what a demfi_pack_planes launch over these nine planes produced in rounds 1-5
struct CfrPack { void* rec; const float* logit; int f32; };
__global__ void cfr_far_kernel(const float* __restrict__ flow01, const float* __restrict__ flow10,
                               const float* __restrict__ tptr, int H, int W, long long* __restrict__ acc,
                               int* __restrict__ tile_flag, int* __restrict__ dbg, CfrBatch bt)
{
    {
        const int q = blockIdx.y;
        flow01 = bofs(flow01, q * bt.f01); flow10 = bofs(flow10, q * bt.f10); tptr = bofs(tptr, q * bt.t);
        acc = bofs(acc, q * bt.acc); tile_flag = bofs(tile_flag, q * bt.acc);
    }
    const int64_t hw = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= 2 * hw) return;
    const int k = (int)(i / hw);
    const int64_t pix = i - k * hw;
    const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
    const float t = *tptr;
    const CfrSrc s = cfr_source(k == 0 ? flow01 : flow10, hw, pix, k == 0 ? t : 1.0f - t);   // fwarp(f01, t f01), fwarp(f10, (1-t) f10)
    if (!s.far_ && !dbg) return;
    long long* a = acc + (int64_t)k * 3 * hw;
    const int tiles_x = (W + CFR_TW - 1) / CFR_TW;
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx) {
        float w;
        const int flat = cfr_target(s, cidx, y, x, H, W, w);
        if (dbg) dbg[((int64_t)k * 4 + cidx) * hw + pix] = flat;
        if (s.far_ && flat >= 0) {
            atomicAdd((unsigned long long*)(a + flat), cfr_fix(s.v0 * w));             // flat_img * flat_weight (fp32), 724
            atomicAdd((unsigned long long*)(a + hw + flat), cfr_fix(s.v1 * w));
            atomicAdd((unsigned long long*)(a + 2 * hw + flat), cfr_fix(w));
            const int r = flat / W, c = flat - r * W;
            tile_flag[(r / CFR_TH) * tiles_x + c / CFR_TW] = 1;                        // benign race: every writer stores 1
        }
    }
}

// One workgroup per TARGET tile: LDS accumulation of the near sources + finish (DeMFInet.py:614-620, one fp32 rounding
// per op) + the far partial sums of flagged tiles.
__global__ __launch_bounds__(NT) void cfr_tile_kernel(const float* __restrict__ flow01, const float* __restrict__ flow10,
                                                      const float* __restrict__ tptr, int H, int W,
                                                      long long* __restrict__ acc, int* __restrict__ tile_flag,
                                                      float* __restrict__ out, CfrBatch bt, CfrPack pk)
{
    __shared__ unsigned long long lacc[6][CFR_TH * CFR_TW];       // [flow k][img0, img1, weight] -> 48 KiB
    {
        const int q = blockIdx.y;
        flow01 = bofs(flow01, q * bt.f01); flow10 = bofs(flow10, q * bt.f10); tptr = bofs(tptr, q * bt.t);
        acc = bofs(acc, q * bt.acc); tile_flag = bofs(tile_flag, q * bt.acc); out = bofs(out, q * bt.out);
        pk.rec = bofs((char*)pk.rec, q * bt.pack); pk.logit = bofs(pk.logit, q * bt.logit);
    }
    const int64_t hw = (int64_t)H * W;
    const int tiles_x = (W + CFR_TW - 1) / CFR_TW;
    const int ntile = tiles_x * ((H + CFR_TH - 1) / CFR_TH);
    // XCD x = blockIdx & 7 owns a contiguous band of tiles: neighbouring tiles share their halos through one L2
    const int per = (ntile + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= ntile || (blockIdx.x >> 3) >= per) return;
    const int ty0 = (tile / tiles_x) * CFR_TH, tx0 = (tile % tiles_x) * CFR_TW;
    for (int i = threadIdx.x; i < 6 * CFR_TH * CFR_TW; i += NT) (&lacc[0][0])[i] = 0ull;
    __syncthreads();
    const float t = *tptr;
    const float omt = 1.0f - t;
    // scanned source window, clipped to the image
    const int sy0 = max(ty0 - CFR_R, 0), sy1 = min(ty0 + CFR_TH + CFR_R, H);
    const int sx0 = max(tx0 - CFR_R, 0), sx1 = min(tx0 + CFR_TW + CFR_R, W);
    const int sw = sx1 - sx0, n = sw * (sy1 - sy0);
    for (int k = 0; k < 2; ++k) {
        const float* fl = k == 0 ? flow01 : flow10;
        const float sc = k == 0 ? t : omt;
        // 4 sources per thread and step: the 8 flow loads are issued together (the loop is latency-bound otherwise)
        for (int i0 = threadIdx.x; i0 < n; i0 += 4 * NT) {
            float v0[4], v1[4];
            int yy[4], xx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * NT, n - 1);
                const int ry = i / sw;
                yy[u] = sy0 + ry;
                xx[u] = sx0 + (i - ry * sw);
                const int64_t pix = (int64_t)yy[u] * W + xx[u];
                v0[u] = fl[pix];
                v1[u] = fl[hw + pix];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u * NT >= n) continue;
                const int y = yy[u], x = xx[u];
                CfrSrc s;
                s.v0 = v0[u]; s.v1 = v1[u];
                s.fy = sc * s.v0; s.fx = sc * s.v1;
                s.x1 = floorf(s.fx); s.y1 = floorf(s.fy);
                if (!(s.x1 >= (float)-CFR_R && s.x1 <= (float)(CFR_R - 1) && s.y1 >= (float)-CFR_R && s.y1 <= (float)(CFR_R - 1)))
                    continue;                                                           // far: cfr_far_kernel's job
                // rows y + x1 + {0,1}, columns x + y1 + {0,1}: skip sources whose 2x2 footprint misses the tile
                const int r0 = y + (int)s.x1 - ty0, c0 = x + (int)s.y1 - tx0;
                if (r0 < -1 || r0 >= CFR_TH || c0 < -1 || c0 >= CFR_TW) continue;
#pragma unroll
                for (int cidx = 0; cidx < 4; ++cidx) {
                    const int r = r0 + (cidx >> 1), c = c0 + (cidx & 1);
                    if (r < 0 || r >= CFR_TH || c < 0 || c >= CFR_TW) continue;
                    if (ty0 + r >= H || tx0 + c >= W) continue;                         // mask (716); >= 0 holds inside a tile
                    const float xs = s.x1 + (float)(cidx >> 1), ys = s.y1 + (float)(cidx & 1);
                    const float dx = s.fx - xs, dy = s.fy - ys;
                    const float w = expf(-(dx * dx + dy * dy));
                    const int li = r * CFR_TW + c;
                    atomicAdd(&lacc[3 * k + 0][li], cfr_fix(s.v0 * w));
                    atomicAdd(&lacc[3 * k + 1][li], cfr_fix(s.v1 * w));
                    atomicAdd(&lacc[3 * k + 2][li], cfr_fix(w));
                }
            }
        }
    }
    __syncthreads();
    const bool flagged = tile_flag[tile] != 0;                   // written by cfr_far_kernel (previous launch on the stream)
    const double inv = 1.0 / FIX;
    for (int li = threadIdx.x; li < CFR_TH * CFR_TW; li += NT) {
        const int r = li / CFR_TW, c = li - r * CFR_TW;
        const int y = ty0 + r, x = tx0 + c;
        if (y >= H || x >= W) continue;
        const int64_t i = (int64_t)y * W + x;
        long long a6[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) a6[q] = (long long)lacc[q][li];
        if (flagged) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const long long g = acc[q * hw + i];
                if (g != 0) { a6[q] += g; acc[q * hw + i] = 0; }  // leave the planes all-zero for the next call
            }
        }
        float f01[2], f10[2];
        f01[0] = (float)((double)a6[0] * inv);
        f01[1] = (float)((double)a6[1] * inv);
        const float n0 = (float)((double)a6[2] * inv);
        f10[0] = (float)((double)a6[3] * inv);
        f10[1] = (float)((double)a6[4] * inv);
        const float n1 = (float)((double)a6[5] * inv);
        const float norm = omt * n0 + t * n1;                                   // 617
        const float m = norm > 0.0f ? 1.0f : 0.0f;                              // 618
        const float ca = (-omt) * t, cb = t * t, cc = omt * omt, cd = t * omt;
        float ftv[4];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            float ft0 = ca * f01[ch] + cb * f10[ch];                            // 614
            float ft1 = cc * f01[ch] - cd * f10[ch];                            // 615
            ft0 = (1.0f - m) * ft0 + m * (ft0 / (norm + (1.0f - m)));           // 619
            ft1 = (1.0f - m) * ft1 + m * (ft1 / (norm + (1.0f - m)));           // 620
            out[(int64_t)ch * hw + i] = ft0;
            out[(int64_t)(2 + ch) * hw + i] = ft1;
            ftv[ch] = ft0; ftv[2 + ch] = ft1;
        }
        if (pk.rec) {
            // [flow_t0, flow_t1 | flow_01, flow_10, logit | zeros]: the same fp32 -> path-dtype conversion as the pack kernel's
            const float v9[9] = {ftv[0], ftv[1], ftv[2], ftv[3], flow01[i], flow01[hw + i], flow10[i], flow10[hw + i], pk.logit[i]};
            if (pk.f32) {
                float* r = (float*)pk.rec + i * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f4_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = 4 * q + j < 9 ? v9[4 * q + j] : 0.0f;
                    *(f4_t*)(r + 4 * q) = o;
                }
            } else {
                half_t* r = (half_t*)pk.rec + i * 16;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    h8_t o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = 8 * q + j < 9 ? (half_t)v9[8 * q + j] : (half_t)0.0f;
                    *(h8_t*)(r + 8 * q) = o;
                }
            }
        }
    }
    __syncthreads();
    if (flagged && threadIdx.x == 0) tile_flag[tile] = 0;
}

// ------------------------------------------------------------------------------------------------------
// bwarp coordinates of one pixel (DeMFInet.py:744-754 + ATen unnormalize) -> SampleMap + validity bit.
__device__ __forceinline__ SampleMap bwarp_map(int x, int y, float fx, float fy, int H, int W, bool& valid)
{
    const float px = (float)x + fx, py = (float)y + fy;
    const float ix = unnormalized_coord(px, (float)max(W - 1, 1), (float)(W - 1));
    const float iy = unnormalized_coord(py, (float)max(H - 1, 1), (float)(H - 1));
    SampleMap m = make_sample_map(ix, iy, H, W);
    const float s = ((m.w[0] + m.w[1]) + m.w[2]) + m.w[3];      // grid_sample of the all-ones image (758-759)
    valid = !(s < 0.999f) && s > 0.0f;                          // masked_fill_ x2 (763-764)
    return m;
}

__device__ __forceinline__ void dbg_store(int* dbg, int which, int64_t hw, int64_t pix, const SampleMap& m, bool valid)
{
    int* d = dbg + (int64_t)which * 3 * hw;
    d[pix] = m.x0;
    d[hw + pix] = m.y0;
    d[2 * hw + pix] = m.inb | (valid ? 16 : 0);
}

// v_fma_mix_f32: d = (f16 half of a) * b + c in fp32 -- folds the fp16->fp32 conversion of a gathered value into the FMA
// (the compiler does not select it here; the blend kernels are VALU-bound, so the saved v_cvt is wall time).
__device__ __forceinline__ float fma_mix_lo(unsigned a, float b, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
#endif
    return d;
}
__device__ __forceinline__ float fma_mix_hi(unsigned a, float b, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
#endif
    return d;
}

template <typename T> struct Vec16;
template <> struct Vec16<half_t> { static constexpr int N = 8; };
template <> struct Vec16<float> { static constexpr int N = 4; };

// Bilinear gather of 16 bytes of channels.  All four corner loads are issued unconditionally (out-of-bounds corners
// read a clamped in-bounds address and carry weight 0, which contributes exactly 0 for finite data), so the 4 (or 8
// with two images) loads of a thread are in flight together instead of being serialised behind divergent branches.
template <typename T>
__device__ __forceinline__ void gather4(const char* base, int64_t sx, int64_t sy, const SampleMap& m, int H, int W, float* o)
{
    constexpr int N = Vec16<T>::N;
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
        if constexpr (sizeof(T) == 2) {
            const h8_t v = __builtin_bit_cast(h8_t, raw[k]);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = o[j] + (float)v[j] * m.w[k];
        } else {
            const f4_t v = __builtin_bit_cast(f4_t, raw[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = o[j] + v[j] * m.w[k];
        }
    }
}

template <typename T>
__device__ __forceinline__ void store16(char* p, const float* v)
{
    if constexpr (sizeof(T) == 2) {
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
        st_global16(p, __builtin_bit_cast(uint4, o));
    } else {
        f4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[j];
        st_global16(p, __builtin_bit_cast(uint4, o));
    }
}

// Fat (NHWC) warp+blend.  The kernel was VALU-bound when every one of the LPP lanes of a pixel redid the coordinate
// round trips (8 IEEE divisions + exp per warp pair): now each wave first computes the maps of 64 consecutive pixels
// with ONE lane per pixel, parks them in a wave-private LDS record, and then walks the pixels LPP lanes at a time
// (16 bytes of channels per lane): broadcast LDS reads, 8 unconditional 16-byte gathers, FMA accumulation.
// Byte strides between the per-t contexts of a batched launch (demfi_batch): context q uses pointer + q * stride.
struct WarpBatch { int nb; int outer; int64_t a, b, o, fa, fb, logit, t, occ, pack; };     // outer: context = blockIdx.y (one launch, contexts one after the other) instead of the innermost loop of a tile
#ifndef DEMFI_WARP_MINW
#define DEMFI_WARP_MINW 1                                        // minimum waves per SIMD the register allocation must allow (A/B builds)
#endif
#ifndef DEMFI_WARP_WGS_DEFAULT
#define DEMFI_WARP_WGS_DEFAULT 0                                 // one tile per workgroup: persistent walks measured 0-14 % SLOWER (profiles/r04_notes.md)
#define DEMFI_WARP_VAR_DEFAULT 1                                 // streaming output stores: -7 % in sequence (the outputs no longer evict F0 / F1 from the Infinity Cache)
#endif

struct WarpRec {                    // 20 dwords per pixel
    int offa[4], offb[4];           // byte offsets of the 4 (clamped) corner records of the two warps
    float wa[4], wb[4];             // corner weights, out-of-bounds and invalid-mask already folded to 0
    float ka, kb, inv_den, den;     // (1-t)*o0, t*(1-o0), 1/den, den
};

template <typename T, int ROWS = 4, bool NTS = false, bool NTL = false>      // ROWS: rows of the tile = waves of the workgroup; NTS: streaming output stores; NTL: streaming gathers (A/B)
__global__ __launch_bounds__(ROWS * 64, DEMFI_WARP_MINW) void warp_blend_fat_kernel(demfi_view A0, const float* __restrict__ fa0, demfi_view B0,
                                      const float* __restrict__ fb0, const float* __restrict__ logit0,
                                      const float* __restrict__ tptr0, demfi_view O0, int lpp_shift, int H, int W,
                                      float* __restrict__ occ_out0, int* __restrict__ dbg, WarpBatch bt)
{
    constexpr int N = Vec16<T>::N;
    __shared__ WarpRec recs[ROWS * 64];                           // 64 records per wave
    const int hw = H * W;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // A workgroup covers a 4-row x 64-pixel tile (wave w = row w), and XCD x = blockIdx & 7 owns a contiguous band of
    // tiles: the two source rows a pixel gathers from are shared with the rows above / below, so that reuse now hits the
    // CU's L1 or the XCD's own L2 instead of being fetched once per XCD (linear 64-pixel spans: rows y and y+1 of one
    // image column sat on different XCDs).
    const int tiles_x = (W + 63) >> 6;
    const int ntile = tiles_x * ((H + ROWS - 1) / ROWS);
    const int per = (ntile + 7) >> 3;
    WarpRec* wr = recs + wave * 64;
    // PERSISTENT tile walk (round 4): gridDim.x / 8 workgroups per XCD walk that XCD's contiguous band of tiles with a stride of
    // the group size, so at any moment the chip works on a narrow window of each band (a few tile rows: the gathered source rows
    // stay in the XCD's L2 and the DRAM pages of F0 / F1 / out are visited once, in order) -- the shape in which a plain streaming
    // kernel of this read : write mix reaches its best rate (tools/microbench/hbm_mix: 512 workgroups 5.6-5.9 TB/s, 2 048: 4.7).
    // With gridDim.x = 8 * per this is the old one-tile-per-workgroup launch.  The flows / logit of the NEXT tile are fetched
    // into registers before this tile's gathers, so phase 1 never waits on HBM again after the first tile.
    const int wg_per_xcd = gridDim.x >> 3;
    const int band0 = (blockIdx.x & 7) * per;
    struct Pre { float ax, ay, bx, by, lg; };
    auto tile_geom = [&](int ti, int& y, int& x0, int& nvalid) {     // false: nothing to do for this wave
        const int tile = band0 + ti;
        if (ti >= per || tile >= ntile) return false;
        const int ty = tile / tiles_x;
        y = ty * ROWS + wave;
        x0 = (tile - ty * tiles_x) << 6;
        nvalid = min(64, W - x0);
        return y < H;
    };
    // Batched launch (bt.nb > 1): the per-t contexts are the INNERMOST loop of the tile -- the source rows the seven time instants of
    // a window gather from are the same neighbourhood of F0 / F1 (flow_t scales with t), so after the first context they come
    // from L1 / the XCD's L2 and the features are fetched from HBM once per window instead of once per time instant.
    const int q_lo = bt.outer ? (int)blockIdx.y : 0, q_hi = bt.outer ? (int)blockIdx.y + 1 : bt.nb;
    for (int q = q_lo; q < q_hi; ++q) {
    demfi_view A = A0, B = B0, O = O0;
    A.ptr = bofs((char*)A0.ptr, q * bt.a); B.ptr = bofs((char*)B0.ptr, q * bt.b); O.ptr = bofs((char*)O0.ptr, q * bt.o);
    const float* __restrict__ fa = bofs(fa0, q * bt.fa);
    const float* __restrict__ fb = bofs(fb0, q * bt.fb);
    const float* __restrict__ logit = bofs(logit0, q * bt.logit);
    const float* __restrict__ tptr = bofs(tptr0, q * bt.t);
    float* __restrict__ occ_out = bofs(occ_out0, q * bt.occ);
    const float t = *tptr;
    auto prefetch = [&](int ti) {
        Pre p = {0.f, 0.f, 0.f, 0.f, 0.f};
        int y, x0, nvalid;
        if (tile_geom(ti, y, x0, nvalid) && lane < nvalid) {
            const int pix = y * W + x0 + lane;
            p.ax = fa[pix]; p.ay = fa[hw + pix]; p.bx = fb[pix]; p.by = fb[hw + pix]; p.lg = logit[pix];
        }
        return p;
    };
    Pre pre = prefetch(blockIdx.x >> 3);
    for (int ti = blockIdx.x >> 3; ti < per; ti += wg_per_xcd) {
    int y, x0, nvalid;
    const bool live = tile_geom(ti, y, x0, nvalid);
    const Pre cur = pre;
    pre = prefetch(ti + wg_per_xcd);                              // in flight during this tile's gathers
    if (!live) continue;                                          // wave-uniform (whole wave past the image / the band)
    const int pix0 = y * W + x0;                                   // first pixel of this wave
    // ---- phase 1: one lane per pixel -------------------------------------------------------------------
    {
        const int pix = pix0 + lane;
        WarpRec r;
        if (lane < nvalid) {
            const int x = x0 + lane;
            bool va, vb;
            const SampleMap ma = bwarp_map(x, y, cur.ax, cur.ay, H, W, va);
            const SampleMap mb = bwarp_map(x, y, cur.bx, cur.by, H, W, vb);
            const float o0 = sigmoidf_(cur.lg);
            const float o1 = 1.0f - o0;
            if (occ_out) occ_out[pix] = o0;
            if (dbg) { dbg_store(dbg, 0, hw, pix, ma, va); dbg_store(dbg, 1, hw, pix, mb, vb); }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r.wa[k] = va ? ma.w[k] : 0.0f;                           // output * mask (766)
                r.wb[k] = vb ? mb.w[k] : 0.0f;
                // out-of-bounds corners carry weight 0: point them at a clamped in-bounds record so the load is legal
                const int ya = min(max(ma.y0 + (k >> 1), 0), H - 1), xa = min(max(ma.x0 + (k & 1), 0), W - 1);
                const int yb = min(max(mb.y0 + (k >> 1), 0), H - 1), xb = min(max(mb.x0 + (k & 1), 0), W - 1);
                r.offa[k] = (int)((ya * A.sy + xa * A.sx) * (int64_t)sizeof(T));
                r.offb[k] = (int)((yb * B.sy + xb * B.sx) * (int64_t)sizeof(T));
            }
            r.ka = (1.0f - t) * o0;
            r.kb = t * o1;
            r.den = r.ka + r.kb;
            r.inv_den = 1.0f / r.den;
            wr[lane] = r;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- phase 2: LPP lanes per pixel, software-pipelined: the 8 gathers of iteration it+1 are in flight while
    // iteration it is blended and stored ----------------------------------------------------------------
    const int lpp = 1 << lpp_shift;
    const int ppi = 64 >> lpp_shift;                              // pixels per iteration
    const int part = lane & (lpp - 1);
    const char* ap = (const char*)A.ptr + part * 16;
    const char* bp = (const char*)B.ptr + part * 16;
    const int psub = lane >> lpp_shift;
    // issue() needs the 8 corner offsets only, finish() the weights / blend factors only: each re-reads its part of the record from
    // LDS (broadcast reads) instead of carrying the whole 20-dword record from one to the other -- 2 x 20 live registers less, which
    // is what lets four waves per SIMD fit (128 registers) without spilling (round 4: the TA is ~70 % busy and the waves wait: more
    // waves in flight is the lever that is left, profiles/r04_notes.md)
    auto issue = [&](int it, uint4 (&ra)[4], uint4 (&rb)[4]) {
        int pl = it * ppi + psub;
        if (pl >= nvalid) pl = 0;                                 // clamp: results of out-of-range pixels are never stored
        const WarpRec& r = wr[pl];
        int oa[4], ob[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { oa[k] = r.offa[k]; ob[k] = r.offb[k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (NTL) {
                ra[k] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(gcp<u4_t>(ap + oa[k])));
                rb[k] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(gcp<u4_t>(bp + ob[k])));
            } else {
                ra[k] = ld_global16(ap + oa[k]);
                rb[k] = ld_global16(bp + ob[k]);
            }
        }
    };
    auto finish = [&](int it, const uint4 (&ra)[4], const uint4 (&rb)[4]) {
        const int pl = it * ppi + psub;
        if (pl >= nvalid) return;
        const int x = x0 + pl;
        WarpRec r;
        {
            const WarpRec& rs = wr[pl];
#pragma unroll
            for (int k = 0; k < 4; ++k) { r.wa[k] = rs.wa[k]; r.wb[k] = rs.wb[k]; }
            r.ka = rs.ka; r.kb = rs.kb; r.inv_den = rs.inv_den; r.den = rs.den;
        }
        float wa[N], wb[N], o[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { wa[j] = 0.0f; wb[j] = 0.0f; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {                             // ATen order nw, ne, sw, se; zero weights add exactly 0
            if constexpr (sizeof(T) == 2) {
                const unsigned pa[4] = {ra[k].x, ra[k].y, ra[k].z, ra[k].w}, pb[4] = {rb[k].x, rb[k].y, rb[k].z, rb[k].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    wa[2 * q] = fma_mix_lo(pa[q], r.wa[k], wa[2 * q]);
                    wa[2 * q + 1] = fma_mix_hi(pa[q], r.wa[k], wa[2 * q + 1]);
                    wb[2 * q] = fma_mix_lo(pb[q], r.wb[k], wb[2 * q]);
                    wb[2 * q + 1] = fma_mix_hi(pb[q], r.wb[k], wb[2 * q + 1]);
                }
            } else {
                const f4_t va = __builtin_bit_cast(f4_t, ra[k]), vb = __builtin_bit_cast(f4_t, rb[k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { wa[j] = wa[j] + va[j] * r.wa[k]; wb[j] = wb[j] + vb[j] * r.wb[k]; }
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if constexpr (sizeof(T) == 2) o[j] = __builtin_fmaf(r.ka, wa[j], r.kb * wb[j]) * r.inv_den;      // Eq.(2), fp16 result
            else o[j] = (r.ka * wa[j] + r.kb * wb[j]) / r.den;                                              // Eq.(2), exact fp32 steps
        }
        char* op = (char*)O.ptr + ((int64_t)y * O.sy + (int64_t)x * O.sx) * sizeof(T) + part * 16;
        if constexpr (NTS) {
            if constexpr (sizeof(T) == 2) {
                h8_t hv;
#pragma unroll
                for (int j = 0; j < 8; ++j) hv[j] = (half_t)o[j];
                st_global16_nt(op, __builtin_bit_cast(uint4, hv));
            } else {
                f4_t fv;
#pragma unroll
                for (int j = 0; j < 4; ++j) fv[j] = o[j];
                st_global16_nt(op, __builtin_bit_cast(uint4, fv));
            }
        } else {
            store16<T>(op, o);
        }
    };
    uint4 ra0[4], rb0[4], ra1[4], rb1[4];
    issue(0, ra0, rb0);
    for (int it = 0; it < lpp; it += 2) {                         // lpp is 8 (fp16) or 16 (fp32): always even
        issue(it + 1, ra1, rb1);
        finish(it, ra0, rb0);
        if (it + 2 < lpp) issue(it + 2, ra0, rb0);
        finish(it + 1, ra1, rb1);
    }
    __builtin_amdgcn_wave_barrier();                              // the records are rewritten for the next tile / context
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }                                                             // tile walk
    }                                                             // contexts
}

// Thin (any strides) warp+blend, one thread per pixel, loops over C channels (C = 3 frames).
// pack8 (optional, C == 3 only): the 8 planes the next layer reads -- [out 0..2 | fa x,y | fb x,y | sigmoid(logit)] -- written
// as ONE NHWC record of the path dtype per pixel (Agg3's per-recursion part, DeMFInet.py:151-155: St_new, rflow_t0, rflow_t1,
// occ), so that no separate plane-packing launch is needed.
__global__ void warp_blend_thin_kernel(demfi_view A, const float* __restrict__ fa, demfi_view B,
                                       const float* __restrict__ fb, const float* __restrict__ logit,
                                       const float* __restrict__ tptr, demfi_view O, int C, int H, int W,
                                       float* __restrict__ occ_out, int* __restrict__ dbg, void* __restrict__ pack8, int pack_f32,
                                       WarpBatch bt)
{
    const int64_t hw = (int64_t)H * W;
    const int64_t pix = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (pix >= hw) return;
    {                                                             // batched launch: blockIdx.y = per-t context
        const int q = blockIdx.y;
        A.ptr = bofs((char*)A.ptr, q * bt.a); B.ptr = bofs((char*)B.ptr, q * bt.b); O.ptr = bofs((char*)O.ptr, q * bt.o);
        fa = bofs(fa, q * bt.fa); fb = bofs(fb, q * bt.fb); logit = bofs(logit, q * bt.logit); tptr = bofs(tptr, q * bt.t);
        occ_out = bofs(occ_out, q * bt.occ); pack8 = bofs((char*)pack8, q * bt.pack);
    }
    const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
    bool va, vb;
    const SampleMap ma = bwarp_map(x, y, fa[pix], fa[hw + pix], H, W, va);
    const SampleMap mb = bwarp_map(x, y, fb[pix], fb[hw + pix], H, W, vb);
    const float t = *tptr;
    const float o0 = sigmoidf_(logit[pix]);
    const float o1 = 1.0f - o0;
    if (occ_out) occ_out[pix] = o0;
    if (dbg) { dbg_store(dbg, 0, hw, pix, ma, va); dbg_store(dbg, 1, hw, pix, mb, vb); }
    const float ka = (1.0f - t) * o0, kb = t * o1;
    const float den = ka + kb;
    float rec[8] = {0.0f, 0.0f, 0.0f, fa[pix], fa[hw + pix], fb[pix], fb[hw + pix], o0};
    // unconditional loads from clamped addresses, out-of-bounds corners select 0 (their weights are 0 too): the sums see the same operands in
    // the same order (a + 0 * 0 == a), without 24 divergent branches per pixel
    int64_t oa[4], ob[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        oa[k] = (int64_t)min(max(ma.y0 + (k >> 1), 0), H - 1) * A.sy + (int64_t)min(max(ma.x0 + (k & 1), 0), W - 1) * A.sx;
        ob[k] = (int64_t)min(max(mb.y0 + (k >> 1), 0), H - 1) * B.sy + (int64_t)min(max(mb.x0 + (k & 1), 0), W - 1) * B.sx;
    }
    // planar fp32 sources (the product's S0' / S1' planes): the two corners of a row are neighbours in memory -- one 8-byte load per row
    // (at a 4-byte boundary) from element xl = clamp(x0, 0, W - 2); corner x0 is element xl (xl + 1 at the right edge), corner x0 + 1 is
    // element xl + 1 (xl at the left edge): 12 gather instructions per pixel instead of 24.  The channel loop stays rolled: unrolled by 3 the
    // kernel loses its occupancy and takes 2.7x as long (measured)
    const bool pair = A.is_f32 && B.is_f32 && A.sx == 1 && B.sx == 1 && W >= 2;
    typedef float F2 __attribute__((ext_vector_type(2), aligned(4)));
    const int xla = min(max(ma.x0, 0), W - 2), xlb = min(max(mb.x0, 0), W - 2);
    if (pair) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            oa[2 * r] = (int64_t)min(max(ma.y0 + r, 0), H - 1) * A.sy + xla;
            ob[2 * r] = (int64_t)min(max(mb.y0 + r, 0), H - 1) * B.sy + xlb;
        }
    }
    for (int c = 0; c < C; ++c) {
        float a = 0.0f, b = 0.0f;
        float ta[4], tb[4];
        if (pair) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const F2 va = *(const DEMFI_GLOBAL F2*)(gcp<float>(A.ptr) + (int64_t)c * A.sc + oa[2 * r]);
                const F2 vb = *(const DEMFI_GLOBAL F2*)(gcp<float>(B.ptr) + (int64_t)c * B.sc + ob[2 * r]);
                ta[2 * r] = ma.x0 == xla ? va.x : va.y;  ta[2 * r + 1] = ma.x0 + 1 == xla ? va.x : va.y;
                tb[2 * r] = mb.x0 == xlb ? vb.x : vb.y;  tb[2 * r + 1] = mb.x0 + 1 == xlb ? vb.x : vb.y;
            }
        } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ta[k] = view_load(A, (int64_t)c * A.sc + oa[k]);
            tb[k] = view_load(B, (int64_t)c * B.sc + ob[k]);
        }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a = a + ((ma.inb & (1 << k)) ? ta[k] : 0.0f) * ma.w[k];
            b = b + ((mb.inb & (1 << k)) ? tb[k] : 0.0f) * mb.w[k];
        }
        a = va ? a : 0.0f;
        b = vb ? b : 0.0f;
        const float v = (ka * a + kb * b) / den;
        view_store(O, (int64_t)c * O.sc + (int64_t)y * O.sy + (int64_t)x * O.sx, v);
        if (c < 3) rec[c] = v;
    }
    if (pack8) {
        if (pack_f32) {
            float* q = (float*)pack8 + pix * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = rec[j];
        } else {
            h8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)rec[j];
            st_global16((char*)pack8 + pix * 16, __builtin_bit_cast(uint4, o));
        }
    }
}

// FGAC sampling at absolute coordinates (DeMFInet.py:413-419, 499-508): one bilinear gather, no validity mask.
template <typename T>
__global__ void fgac_gather_kernel(demfi_view S, const float* __restrict__ flow, demfi_view O, int lpp_shift, int H,
                                   int W, int* __restrict__ dbg)
{
    constexpr int N = Vec16<T>::N;
    const int64_t hw = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int64_t pix = i >> lpp_shift;
    if (pix >= hw) return;
    const int part = (int)(i & ((1 << lpp_shift) - 1));
    const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
    const float ix = unnormalized_coord(flow[pix], (float)(W - 1), (float)(W - 1));
    const float iy = unnormalized_coord(flow[hw + pix], (float)(H - 1), (float)(H - 1));
    const SampleMap m = make_sample_map(ix, iy, H, W);
    if (dbg && part == 0) dbg_store(dbg, 0, hw, pix, m, true);
    float r[N];
    gather4<T>((const char*)S.ptr + part * 16, S.sx * sizeof(T), S.sy * sizeof(T), m, H, W, r);
    store16<T>((char*)O.ptr + ((int64_t)y * O.sy + (int64_t)x * O.sx) * sizeof(T) + part * 16, r);
}

// Eq.(4): out = w*source + (1-w)*e  (DeMFInet.py:452)
template <typename T>
__global__ void gate_blend_kernel(const float* __restrict__ w, demfi_view S, demfi_view E, demfi_view O, int lpp_shift,
                                  int H, int W)
{
    constexpr int N = Vec16<T>::N;
    const int64_t hw = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int64_t pix = i >> lpp_shift;
    if (pix >= hw) return;
    const int part = (int)(i & ((1 << lpp_shift) - 1));
    const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
    const float g = w[pix];
    const uint4 sr = ld_global16((const char*)S.ptr + ((int64_t)y * S.sy + (int64_t)x * S.sx) * sizeof(T) + part * 16);
    const uint4 er = ld_global16((const char*)E.ptr + ((int64_t)y * E.sy + (int64_t)x * E.sx) * sizeof(T) + part * 16);
    float r[N];
    if constexpr (sizeof(T) == 2) {
        const h8_t s = __builtin_bit_cast(h8_t, sr), e = __builtin_bit_cast(h8_t, er);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = g * (float)s[j] + (1.0f - g) * (float)e[j];
    } else {
        const f4_t s = __builtin_bit_cast(f4_t, sr), e = __builtin_bit_cast(f4_t, er);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = g * s[j] + (1.0f - g) * e[j];
    }
    store16<T>((char*)O.ptr + ((int64_t)y * O.sy + (int64_t)x * O.sx) * sizeof(T) + part * 16, r);
}

// Gather up to 32 planar fp32 channels (flows, logits, frames) into an NHWC slice of the path dtype, so that the
// consuming convolution stages them with 16-byte vector loads instead of element-wise "thin" loads.
struct PackArgs { const float* plane[32]; };
struct PackBatch { int64_t plane[32]; int64_t dst; };          // byte strides between per-t contexts (blockIdx.y)

template <typename T>
__global__ void pack_planes_kernel(PackArgs a, int nch8, T* __restrict__ dst, int64_t dst_sx, int hw, PackBatch bt)
{
    constexpr int G = 16 / sizeof(T);                  // channels per 16-byte store
    const int i = blockIdx.x * NT + threadIdx.x;
    const int ngrp = nch8 * 8 / G;
    if (i >= hw * ngrp) return;
    const int q = blockIdx.y;                           // batched launch: per-t context
    const int g = i / hw, pix = i - g * hw;            // pixel fastest: plane reads coalesce
    float v[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const float* p = bofs(a.plane[g * G + j], q * bt.plane[g * G + j]);
        v[j] = p ? p[pix] : 0.0f;
    }
    store16<T>((char*)(dst + (int64_t)pix * dst_sx + g * G) + q * bt.dst, v);
}

// ---- uint8 frame I/O of the boundary caller --------------------------------------------
// Input side: cv2.imread BGR uint8 [h,w,3] x 4 frames -> RGBframes_np2Tensor (utils.py:224-238): (u/255 - 0.5)*2 in
// fp32, layout [C,T,H,W] -- fused with the harness' bottom/right reflect padding (utils.py:1363).
struct U8Frames { const unsigned char* f[4]; };

__global__ void u8_to_window_kernel(U8Frames fr, float* __restrict__ x, int h, int w, int H, int W)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= 4 * H * W) return;
    const int X = i % W, Y = (i / W) % H, f = i / (W * H);
    const int sx = X < w ? X : 2 * (w - 1) - X;
    const int sy = Y < h ? Y : 2 * (h - 1) - Y;
    const unsigned char* p = fr.f[f] + ((int64_t)sy * w + sx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = (float)p[c] / 255.0f;
        v = v - 0.5f;
        x[((int64_t)(c * 4 + f) * H + Y) * W + X] = v * 2.0f;
    }
}

// u8_to_window fused with the first loads of the network: one thread per half-resolution pixel reads the 2x2 block of the 4
// frames once and writes x (fp32 planes), the space-to-depth record of FF_RDB (48 channels: (frame*3 + c)*4 + ry*2 + rx,
// DeMFInet.py:311-316) and the overlay mean of B0, B1 (DeMFInet.py:178).
template <typename T>
__global__ void u8_ingest_kernel(U8Frames fr, float* __restrict__ x, T* __restrict__ s2d, float* __restrict__ ov, int h, int w,
                                 int H, int W)
{
    const int H2 = H >> 1, W2 = W >> 1;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H2 * W2) return;
    const int x2 = i % W2, y2 = i / W2;
    T rec[48];
    float b01[2][3][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int Y = 2 * y2 + (q >> 1), X = 2 * x2 + (q & 1);
            const int sx = X < w ? X : 2 * (w - 1) - X;
            const int sy = Y < h ? Y : 2 * (h - 1) - Y;
            const unsigned char* p = fr.f[f] + ((int64_t)sy * w + sx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = (float)p[c] / 255.0f;
                v = v - 0.5f;
                v = v * 2.0f;
                x[((int64_t)(c * 4 + f) * H + Y) * W + X] = v;
                rec[(f * 3 + c) * 4 + q] = (T)v;
                if (f < 2) b01[f][c][q] = v;
            }
        }
    }
    T* o = s2d + (int64_t)i * 48;
#pragma unroll
    for (int k = 0; k < 48 * (int)sizeof(T) / 16; ++k) st_global16((char*)o + k * 16, ((const uint4*)rec)[k]);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ov[((int64_t)c * H + 2 * y2 + (q >> 1)) * W + 2 * x2 + (q & 1)] = (b01[0][c][q] + b01[1][c][q]) / 2.0f;
}

// One uint8 frame -> planar fp32 [3,h,w] with the loader's arithmetic (targets of the on-GPU evaluation).
__global__ void u8_to_planar_kernel(const unsigned char* __restrict__ f, float* __restrict__ out, int hw)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= hw) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = (float)f[(int64_t)i * 3 + c] / 255.0f;
        v = v - 0.5f;
        out[(int64_t)c * hw + i] = v * 2.0f;
    }
}

// Output side: denorm255_np (utils.py:718-721) on the float64 copy of the fp32 frame, then .astype(np.uint8)
// truncation (main.py:1165-1178), cropped to h x w, HWC.
__global__ void frame_to_u8_kernel(const float* __restrict__ fr, unsigned char* __restrict__ out, int h, int w, int H, int W)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= h * w) return;
    const int X = i % w, Y = i / w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = ((double)fr[((int64_t)c * H + Y) * W + X] + 1.0) / 2.0;
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        out[(int64_t)i * 3 + c] = (unsigned char)(v * 255.0);
    }
}

// A fat view usable by the 16-byte-per-lane kernels: NHWC (sc == 1), C*elt a power-of-two multiple of 16 B.
int fat_lpp_shift(const demfi_view* v, int C, const char* who, int* is_f32)
{
    if (!v || !v->ptr || v->sc != 1) return demfi_set_error(DEMFI_ERR_ARG, "%s: view is not NHWC", who);
    const int bytes = C * (v->is_f32 ? 4 : 2);
    const int lpp = bytes / 16;
    if (bytes % 16 || lpp < 1 || (lpp & (lpp - 1))) return demfi_set_error(DEMFI_ERR_ARG, "%s: C=%d not vectorisable", who, C);
    *is_f32 = v->is_f32;
    int s = 0;
    while ((1 << s) < lpp) ++s;
    return s;
}

}  // namespace

extern "C" int demfi_space_to_depth(const float* x, void* out, int dtype, int H, int W, void* stream)
{
    if (!x || !out || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return demfi_set_error(DEMFI_ERR_ARG, "demfi_space_to_depth: bad args");
    const int64_t n = (int64_t)12 * (H / 2) * (W / 2);
    if (dtype == DEMFI_F16)
        hipLaunchKernelGGL(s2d_kernel<half_t>, dim3(blocks_for(n)), dim3(NT), 0, (hipStream_t)stream, x, (half_t*)out, H, W);
    else if (dtype == DEMFI_F32)
        hipLaunchKernelGGL(s2d_kernel<float>, dim3(blocks_for(n)), dim3(NT), 0, (hipStream_t)stream, x, (float*)out, H, W);
    else
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_space_to_depth: dtype");
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_reflect_pad(const float* x, float* out, int planes, int h, int w, int H, int W, void* stream)
{
    if (!x || !out || planes <= 0 || h < 2 || w < 2 || H < h || W < w || H - h >= h || W - w >= w)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_reflect_pad: bad sizes %dx%d -> %dx%d", h, w, H, W);
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(blocks_for((int64_t)planes * H * W)), dim3(NT), 0, (hipStream_t)stream,
                       x, out, planes, h, w, H, W);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_overlay_mean(const float* x, float* out, int H, int W, void* stream)
{
    if (!x || !out || H <= 0 || W <= 0) return demfi_set_error(DEMFI_ERR_ARG, "demfi_overlay_mean: bad args");
    hipLaunchKernelGGL(overlay_kernel, dim3(blocks_for((int64_t)3 * H * W)), dim3(NT), 0, (hipStream_t)stream, x, out,
                       (int64_t)H * W);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int64_t demfi_cfr_workspace_bytes(int H, int W)
{
    if (H <= 0 || W <= 0) return 0;
    const int64_t tiles = (int64_t)((W + CFR_TW - 1) / CFR_TW) * ((H + CFR_TH - 1) / CFR_TH);
    return 6 * (int64_t)H * W * 8 + ((tiles * 4 + 255) & ~255ll);
}

// The accumulators are self-cleaning, which holds only for launches that run to completion: after an aborted launch (a
// failed hipGraph replay, a device reset survived by the allocation) the caller re-zeroes them here.
extern "C" int demfi_cfr_reset(int64_t* acc, int H, int W, void* stream)
{
    if (!acc || H <= 0 || W <= 0) return demfi_set_error(DEMFI_ERR_ARG, "demfi_cfr_reset: bad args");
    DEMFI_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)demfi_cfr_workspace_bytes(H, W), (hipStream_t)stream));
    return DEMFI_OK;
}

static int cfr_impl(const float* flow01, const float* flow10, const float* t, int H, int W, int64_t* acc, float* out, int32_t* dbg_idx,
                    const demfi_batch* bt, void* stream, const float* logit = nullptr, void* pack16 = nullptr, int pack_dtype = DEMFI_F16)
{
    if (!flow01 || !flow10 || !t || !acc || !out || H <= 0 || W <= 0 || (int64_t)H * W >= (1ll << 31))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_cfr_flow_align: bad args");
    const int nb = bt && bt->nb > 1 ? bt->nb : 1;
    if (nb > 1 && dbg_idx) return demfi_set_error(DEMFI_ERR_ARG, "demfi_cfr_flow_align_batched: no debug maps in a batched launch");
    if ((pack16 != nullptr) != (logit != nullptr) || (pack_dtype != DEMFI_F16 && pack_dtype != DEMFI_F32))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_cfr_flow_align_pack: the packed record needs the occlusion-logit plane (and a path dtype)");
    CfrBatch cb = {0, 0, 0, 0, 0, 0, 0};
    if (nb > 1) { cb.f01 = bt->p[0]; cb.f10 = bt->p[1]; cb.t = bt->t; cb.acc = bt->p[2]; cb.out = bt->p[3]; cb.logit = bt->p[4]; cb.pack = bt->p[5]; }
    const CfrPack pk = {pack16, logit, pack_dtype == DEMFI_F32 ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const int64_t hw = (int64_t)H * W;
    int* tile_flag = (int*)(acc + 6 * hw);                       // behind the six int64 planes (demfi_cfr_workspace_bytes)
    hipLaunchKernelGGL(cfr_far_kernel, dim3(blocks_for(2 * hw), nb), dim3(NT), 0, st, flow01, flow10, t, H, W,
                       (long long*)acc, tile_flag, dbg_idx, cb);
    const int ntile = ((W + CFR_TW - 1) / CFR_TW) * ((H + CFR_TH - 1) / CFR_TH);
    hipLaunchKernelGGL(cfr_tile_kernel, dim3(8 * ((ntile + 7) / 8), nb), dim3(NT), 0, st, flow01, flow10, t, H, W,
                       (long long*)acc, tile_flag, out, cb, pk);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_cfr_flow_align_pack(const float* flow01, const float* flow10, const float* logit, const float* t, int H, int W, int64_t* acc,
                                         float* out, void* pack16, int pack_dtype, const demfi_batch* bt, void* stream)
{
    if (bt && (bt->nb < 1 || bt->nb > 64)) return demfi_set_error(DEMFI_ERR_ARG, "demfi_cfr_flow_align_pack: batch description");
    return cfr_impl(flow01, flow10, t, H, W, acc, out, nullptr, bt, stream, logit, pack16, pack_dtype);
}

extern "C" int demfi_cfr_flow_align(const float* flow01, const float* flow10, const float* t, int H, int W,
                                    int64_t* acc, float* out, int32_t* dbg_idx, void* stream)
{
    return cfr_impl(flow01, flow10, t, H, W, acc, out, dbg_idx, nullptr, stream);
}

extern "C" int demfi_cfr_flow_align_batched(const float* flow01, const float* flow10, const float* t, int H, int W, int64_t* acc,
                                            float* out, const demfi_batch* bt, void* stream)
{
    if (!bt || bt->nb < 1 || bt->nb > 64) return demfi_set_error(DEMFI_ERR_ARG, "demfi_cfr_flow_align_batched: batch description");
    return cfr_impl(flow01, flow10, t, H, W, acc, out, nullptr, bt, stream);
}

static int warp_blend_impl(const demfi_view* A, const float* fa, const demfi_view* B, const float* fb, const float* logit,
                           const float* t, const demfi_view* out, int C, int H, int W, float* occ_out, int32_t* dbg_maps,
                           void* pack8, int pack_dtype, void* stream, const demfi_batch* bt = nullptr)
{
    if (!A || !B || !out || !A->ptr || !B->ptr || !out->ptr || !fa || !fb || !logit || !t || C <= 0 || H <= 0 || W <= 0)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_warp_blend: bad args");
    WarpBatch wb = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (bt && bt->nb > 1) {
        wb.outer = bt->_pad == 1 ? 1 : 0;
        if (dbg_maps) return demfi_set_error(DEMFI_ERR_ARG, "demfi_warp_blend_batched: no debug maps in a batched launch");
        wb.nb = bt->nb; wb.a = bt->a; wb.b = bt->b; wb.o = bt->o; wb.fa = bt->p[0]; wb.fb = bt->p[1]; wb.logit = bt->p[2]; wb.t = bt->t;
        wb.occ = bt->p[3]; wb.pack = bt->p[4];
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t hw = (int64_t)H * W;
    const bool fat = A->sc == 1 && B->sc == 1 && out->sc == 1 && A->is_f32 == B->is_f32 && A->is_f32 == out->is_f32
                     && (C * (A->is_f32 ? 4 : 2)) % 16 == 0;
    if (fat && pack8) return demfi_set_error(DEMFI_ERR_ARG, "demfi_warp_blend_pack: planar views expected");
    if (fat) {
        int f32 = 0;
        const int sh = fat_lpp_shift(A, C, "demfi_warp_blend", &f32);
        if (sh < 0) return sh;
        // the kernel keeps the corner byte offsets of A / B as 32-bit ints (WarpRec): refuse images they cannot address
        const int64_t elt = f32 ? 4 : 2;
        if (((int64_t)(H - 1) * A->sy + (int64_t)(W - 1) * A->sx + C) * elt >= ((int64_t)1 << 31) ||
            ((int64_t)(H - 1) * B->sy + (int64_t)(W - 1) * B->sx + C) * elt >= ((int64_t)1 << 31))
            return demfi_set_error(DEMFI_ERR_ARG, "demfi_warp_blend: image spans >= 2^31 bytes (32-bit corner offsets)");
        // ROWS-row x 64-pixel tiles, 8 XCD bands of ceil(ntile / 8) tiles each
        static const int var = getenv("DEMFI_WARP_VAR") ? atoi(getenv("DEMFI_WARP_VAR")) : DEMFI_WARP_VAR_DEFAULT;   // probe switch (tools/conv_probe.py warp): bit 0 = non-temporal output stores, bit 1 = 8-row tiles
        static const int wgs = getenv("DEMFI_WARP_WGS") ? atoi(getenv("DEMFI_WARP_WGS")) : DEMFI_WARP_WGS_DEFAULT;   // persistent workgroups per launch (multiple of 8; 0 = one tile per workgroup)
        const int rows = (var & 2) ? 8 : 4;
        unsigned nblk = 8u * (unsigned)((((W + 63) / 64) * ((H + rows - 1) / rows) + 7) / 8);
        if (wgs >= 8 && (unsigned)(wgs & ~7) < nblk) nblk = (unsigned)(wgs & ~7);
#define DEMFI_WARP_LAUNCH(TT, R, N, NL)                                                                               \
        hipLaunchKernelGGL((warp_blend_fat_kernel<TT, R, N, NL>), dim3(nblk, wb.outer ? wb.nb : 1), dim3(R * 64), 0, st, *A, fa, *B, fb, logit, t, *out, sh, H, W, \
                           occ_out, dbg_maps, wb)
        if (f32) {
            if ((var & 3) == 0) DEMFI_WARP_LAUNCH(float, 4, false, false); else if ((var & 3) == 1) DEMFI_WARP_LAUNCH(float, 4, true, false);
            else if ((var & 3) == 2) DEMFI_WARP_LAUNCH(float, 8, false, false); else DEMFI_WARP_LAUNCH(float, 8, true, false);
        } else {
            if (var == 0) DEMFI_WARP_LAUNCH(half_t, 4, false, false); else if (var == 1) DEMFI_WARP_LAUNCH(half_t, 4, true, false);
            else if (var == 2) DEMFI_WARP_LAUNCH(half_t, 8, false, false); else if (var == 3) DEMFI_WARP_LAUNCH(half_t, 8, true, false);
            else if (var == 4) DEMFI_WARP_LAUNCH(half_t, 4, false, true); else DEMFI_WARP_LAUNCH(half_t, 4, true, true);      // 4, 5: streaming gathers
        }
#undef DEMFI_WARP_LAUNCH
    } else {
        if (pack8 && C != 3) return demfi_set_error(DEMFI_ERR_ARG, "demfi_warp_blend_pack: the packed record is defined for C == 3");
        hipLaunchKernelGGL(warp_blend_thin_kernel, dim3(blocks_for(hw), wb.nb), dim3(NT), 0, st, *A, fa, *B, fb, logit, t, *out, C,
                           H, W, occ_out, dbg_maps, pack8, pack_dtype == DEMFI_F32 ? 1 : 0, wb);
    }
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_warp_blend(const demfi_view* A, const float* fa, const demfi_view* B, const float* fb,
                                const float* logit, const float* t, const demfi_view* out, int C, int H, int W,
                                float* occ_out, int32_t* dbg_maps, void* stream)
{
    return warp_blend_impl(A, fa, B, fb, logit, t, out, C, H, W, occ_out, dbg_maps, nullptr, 0, stream);
}

extern "C" int demfi_warp_blend_pack(const demfi_view* A, const float* fa, const demfi_view* B, const float* fb,
                                     const float* logit, const float* t, const demfi_view* out, int H, int W, float* occ_out,
                                     void* pack8, int pack_dtype, void* stream)
{
    if (!pack8 || (pack_dtype != DEMFI_F16 && pack_dtype != DEMFI_F32) || !A || A->sc == 1)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_warp_blend_pack: planar 3-channel views and a pack buffer expected");
    return warp_blend_impl(A, fa, B, fb, logit, t, out, 3, H, W, occ_out, nullptr, pack8, pack_dtype, stream);
}

extern "C" int demfi_warp_blend_batched(const demfi_view* A, const float* fa, const demfi_view* B, const float* fb, const float* logit,
                                        const float* t, const demfi_view* out, int C, int H, int W, float* occ_out, void* pack8,
                                        int pack_dtype, const demfi_batch* bt, void* stream)
{
    if (!bt || bt->nb < 1 || bt->nb > 64) return demfi_set_error(DEMFI_ERR_ARG, "demfi_warp_blend_batched: batch description");
    if (pack8 && (pack_dtype != DEMFI_F16 && pack_dtype != DEMFI_F32)) return demfi_set_error(DEMFI_ERR_ARG, "demfi_warp_blend_batched: pack dtype");
    return warp_blend_impl(A, fa, B, fb, logit, t, out, C, H, W, occ_out, nullptr, pack8, pack_dtype, stream, bt);
}

extern "C" int demfi_fgac_gather(const demfi_view* src, const float* flow, const demfi_view* out, int C, int H, int W,
                                 int32_t* dbg_maps, void* stream)
{
    if (!src || !out || !flow || H <= 1 || W <= 1) return demfi_set_error(DEMFI_ERR_ARG, "demfi_fgac_gather: bad args");
    int f32 = 0, f32o = 0;
    const int sh = fat_lpp_shift(src, C, "demfi_fgac_gather", &f32);
    if (sh < 0) return sh;
    if (fat_lpp_shift(out, C, "demfi_fgac_gather", &f32o) != sh || f32o != f32)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_fgac_gather: src/out type mismatch");
    const int64_t n = ((int64_t)H * W) << sh;
    hipStream_t st = (hipStream_t)stream;
    if (f32)
        hipLaunchKernelGGL(fgac_gather_kernel<float>, dim3(blocks_for(n)), dim3(NT), 0, st, *src, flow, *out, sh, H, W, dbg_maps);
    else
        hipLaunchKernelGGL(fgac_gather_kernel<half_t>, dim3(blocks_for(n)), dim3(NT), 0, st, *src, flow, *out, sh, H, W, dbg_maps);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_gate_blend(const float* w, const demfi_view* source, const demfi_view* e, const demfi_view* out,
                                int C, int H, int W, void* stream)
{
    if (!w || !source || !e || !out) return demfi_set_error(DEMFI_ERR_ARG, "demfi_gate_blend: bad args");
    int f32 = 0, f2 = 0, f3 = 0;
    const int sh = fat_lpp_shift(source, C, "demfi_gate_blend", &f32);
    if (sh < 0) return sh;
    if (fat_lpp_shift(e, C, "demfi_gate_blend", &f2) != sh || fat_lpp_shift(out, C, "demfi_gate_blend", &f3) != sh || f2 != f32 || f3 != f32)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_gate_blend: view type mismatch");
    const int64_t n = ((int64_t)H * W) << sh;
    hipStream_t st = (hipStream_t)stream;
    if (f32)
        hipLaunchKernelGGL(gate_blend_kernel<float>, dim3(blocks_for(n)), dim3(NT), 0, st, w, *source, *e, *out, sh, H, W);
    else
        hipLaunchKernelGGL(gate_blend_kernel<half_t>, dim3(blocks_for(n)), dim3(NT), 0, st, w, *source, *e, *out, sh, H, W);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

static int pack_impl(const float* const* planes, int nch, void* dst, int dtype, int64_t dst_pix_stride, int H, int W,
                     const demfi_batch* bt, void* stream)
{
    if (!planes || !dst || nch <= 0 || nch > 32 || nch % 8 || H <= 0 || W <= 0)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_pack_planes: nch=%d must be a multiple of 8, <= 32", nch);
    PackArgs a;
    PackBatch pb;
    const int nb = bt && bt->nb > 1 ? bt->nb : 1;
    for (int i = 0; i < 32; ++i) { a.plane[i] = i < nch ? planes[i] : nullptr; pb.plane[i] = nb > 1 ? bt->p[i] : 0; }
    pb.dst = nb > 1 ? bt->o : 0;
    const int hw = H * W;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DEMFI_F16)
        hipLaunchKernelGGL(pack_planes_kernel<half_t>, dim3(blocks_for((int64_t)hw * nch / 8), nb), dim3(NT), 0, st, a, nch / 8,
                           (half_t*)dst, dst_pix_stride, hw, pb);
    else if (dtype == DEMFI_F32)
        hipLaunchKernelGGL(pack_planes_kernel<float>, dim3(blocks_for((int64_t)hw * nch / 4), nb), dim3(NT), 0, st, a, nch / 8,
                           (float*)dst, dst_pix_stride, hw, pb);
    else
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_pack_planes: dtype");
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_pack_planes(const float* const* planes, int nch, void* dst, int dtype, int64_t dst_pix_stride,
                                 int H, int W, void* stream)
{
    return pack_impl(planes, nch, dst, dtype, dst_pix_stride, H, W, nullptr, stream);
}

extern "C" int demfi_pack_planes_batched(const float* const* planes, int nch, void* dst, int dtype, int64_t dst_pix_stride, int H, int W,
                                         const demfi_batch* bt, void* stream)
{
    if (!bt || bt->nb < 1 || bt->nb > 64) return demfi_set_error(DEMFI_ERR_ARG, "demfi_pack_planes_batched: batch description");
    return pack_impl(planes, nch, dst, dtype, dst_pix_stride, H, W, bt, stream);
}

extern "C" int demfi_u8_to_window(const uint8_t* const* frames, int h, int w, float* x, int H, int W, void* stream)
{
    if (!frames || !x || h < 2 || w < 2 || H < h || W < w || H - h >= h || W - w >= w)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_u8_to_window: bad sizes %dx%d -> %dx%d", h, w, H, W);
    U8Frames fr;
    for (int i = 0; i < 4; ++i) {
        if (!frames[i]) return demfi_set_error(DEMFI_ERR_ARG, "demfi_u8_to_window: frame %d is NULL", i);
        fr.f[i] = frames[i];
    }
    hipLaunchKernelGGL(u8_to_window_kernel, dim3(blocks_for((int64_t)4 * H * W)), dim3(NT), 0, (hipStream_t)stream, fr, x, h, w, H, W);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_u8_ingest(const uint8_t* const* frames, int h, int w, float* x, void* s2d, float* overlay, int dtype, int H,
                               int W, void* stream)
{
    if (!frames || !x || !s2d || !overlay || h < 2 || w < 2 || H < h || W < w || H - h >= h || W - w >= w || (H & 1) || (W & 1))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_u8_ingest: bad sizes %dx%d -> %dx%d", h, w, H, W);
    U8Frames fr;
    for (int i = 0; i < 4; ++i) {
        if (!frames[i]) return demfi_set_error(DEMFI_ERR_ARG, "demfi_u8_ingest: frame %d is NULL", i);
        fr.f[i] = frames[i];
    }
    const int64_t n = (int64_t)(H / 2) * (W / 2);
    if (dtype == DEMFI_F16)
        hipLaunchKernelGGL(u8_ingest_kernel<half_t>, dim3(blocks_for(n)), dim3(NT), 0, (hipStream_t)stream, fr, x, (half_t*)s2d, overlay, h, w, H, W);
    else if (dtype == DEMFI_F32)
        hipLaunchKernelGGL(u8_ingest_kernel<float>, dim3(blocks_for(n)), dim3(NT), 0, (hipStream_t)stream, fr, x, (float*)s2d, overlay, h, w, H, W);
    else
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_u8_ingest: dtype");
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_u8_to_planar(const uint8_t* frame, int h, int w, float* out, void* stream)
{
    if (!frame || !out || h <= 0 || w <= 0) return demfi_set_error(DEMFI_ERR_ARG, "demfi_u8_to_planar: bad args");
    hipLaunchKernelGGL(u8_to_planar_kernel, dim3(blocks_for((int64_t)h * w)), dim3(NT), 0, (hipStream_t)stream, frame, out, h * w);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

extern "C" int demfi_frame_to_u8(const float* frame, uint8_t* out, int h, int w, int H, int W, void* stream)
{
    if (!frame || !out || h <= 0 || w <= 0 || H < h || W < w) return demfi_set_error(DEMFI_ERR_ARG, "demfi_frame_to_u8: bad args");
    hipLaunchKernelGGL(frame_to_u8_kernel, dim3(blocks_for((int64_t)h * w)), dim3(NT), 0, (hipStream_t)stream, frame, out, h, w, H, W);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}
