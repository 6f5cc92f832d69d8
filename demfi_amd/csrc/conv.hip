// Implicit-GEMM convolution for gfx950 matrix cores (MFMA), DeMFI-Net_rb forward path.
//
// Replaces every nn.Conv2d / nn.Conv3d(1,k,k) call site of the reference together with the
// torch.cat / PixelShuffle / UpsamplingNearest2d / activation / residual / GRU-gate ops around them
// (DeMFInet.py:209-231, 324-378, 575-584, 30-44, 800-868).
//
// Mapping (one workgroup = 256 threads = 4 wave64):
//   output tile      : 8 rows x 32 columns of pixels, NCO x 32 output channels
//   wave w           : rows 2w, 2w+1 (two 32-pixel MFMA column blocks) x all NCO cout subtiles
//   MFMA             : D[cout][pixel] += W[cout][k] * X[k][pixel]
//                      fp16: v_mfma_f32_32x32x16_f16 (A = 8 packed weights / lane, B = 8 channels of one pixel)
//                      fp32: 4 x v_mfma_f32_32x32x2_f32 on the same 16-byte operands (exact fp32)
//   input staging    : per chunk (<= 128 B of channels per pixel) the haloed input tile
//                      [(8-1)*s+kh] x [(32-1)*s+kw] pixels is gathered from up to several source views into
//                      LDS (record stride rec+16 B => conflict-free ds_read_b128 across 16 consecutive
//                      pixels); every filter tap then reads its B fragments from LDS (kh*kw-fold reuse).
//   weights          : pre-packed in A-fragment order (demfi_pack_conv_weights); one tap's fragments are
//                      DMA'd global->LDS (global_load_lds, 1 KiB per wave-instruction) into a 2-deep ring one
//                      tap ahead of the MFMAs and shared by the 4 waves (one barrier per tap).
//   epilogue         : bias + residual + activation / GRU gate math on the accumulators, routed per
//                      8-cout octet to strided destination views (NHWC slices, planar fp32, PixelShuffle).
//   grid             : x = spatial tiles (XCD-aware: each XCD's L2 gets a contiguous band of tiles so that
//                      halos are shared inside one L2), y = cout blocks, z = batch.
#include "common.h"
#include <type_traits>

namespace {

constexpr int TH = 8;
constexpr int TW = 32;
constexpr int NT = 256;
constexpr int REC_PAD = 16;

template <typename T> struct Mma;

template <> struct Mma<half_t> {
    static __device__ __forceinline__ void run(f16x_t& acc, const uint4& a, const uint4& b)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b),
                                                     acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(f16x_t& acc, const uint4& a, const uint4& b)
    {
        f4_t fa = __builtin_bit_cast(f4_t, a), fb = __builtin_bit_cast(f4_t, b);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc, 0, 0, 0);
    }
};

// Compile-time loop: the accumulator arrays must only ever be indexed by constants (runtime-indexed
// ext_vector arrays go to scratch), and '#pragma unroll' is refused on the large epilogue body.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int STAGE_LD = 36;          // floats per staged pixel row: 32 couts + 4 pad (144 B, conflict-light b128)

template <typename T> __device__ __forceinline__ void load8(const T* p, float* o);
template <> __device__ __forceinline__ void load8<half_t>(const half_t* p, float* o)
{
    const h8_t v = *(const h8_t*)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float* o)
{
    const f4_t a = *(const f4_t*)p, b = *(const f4_t*)(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v);
template <> __device__ __forceinline__ void store8<half_t>(half_t* p, const float* v)
{
    h8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
    *(h8_t*)p = o;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v)
{
    f4_t a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
    *(f4_t*)p = a;
    *(f4_t*)(p + 4) = b;
}

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act == DEMFI_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == DEMFI_ACT_TANH) return tanhf(v);
    if (act == DEMFI_ACT_SIGMOID) return sigmoidf_(v);
    return v;
}

// second launch_bounds argument = minimum waves per SIMD: 2-3 resident workgroups per CU let one workgroup's
// tile staging overlap another's MFMA phase.
template <typename T, int NCO>
__global__ __launch_bounds__(NT, (NCO <= 1 ? 4 : (NCO == 2 ? 3 : 2))) void conv_kernel(const demfi_conv* __restrict__ d)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int lx = lane & 31;

    const int H = d->H, W = d->W, inH = d->inH, inW = d->inW;
    const int kh = d->kh, kw = d->kw, stride = d->stride;
    const int tiles_x = (W + TW - 1) / TW;
    const int tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y;

    // XCD-aware bijective remap: workgroup b runs on XCD b % 8 (observed); give each XCD a contiguous band.
    int tile;
    {
        const int bid = blockIdx.x;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ty = tile / tiles_x;
    const int tx = tile - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int cblk = blockIdx.y;
    const int bimg = blockIdx.z;

    const int LW = (TW - 1) * stride + kw;
    const int LH = (TH - 1) * stride + kh;
    const int NP = LH * LW;
    const int rec = d->rec_bytes + REC_PAD;
    const int iy0 = oy0 * stride - d->pad_y;
    const int ix0 = ox0 * stride - d->pad_x;
    const uint32_t lw_magic = d->lw_magic;
    constexpr int ESZ = sizeof(T);

    f16x_t acc[NCO][2];
#pragma unroll
    for (int s = 0; s < NCO; ++s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[s][0][i] = 0.0f; acc[s][1][i] = 0.0f; }
    }

    const uint4* __restrict__ wbase = (const uint4*)d->wpack + (int64_t)cblk * d->w_blk_stride;
    const int ntaps = kh * kw;
    const int wbuf_bytes = (d->rec_bytes >> 5) * NCO * 1024;        // one tap: nks_max x NCO fragments of 1 KiB
    char* const wlds = smem + ((NP * rec + 1023) & ~1023);          // weight ring (2 taps) behind the input tile
    // LDS-DMA of one tap's A fragments: piece i (1 KiB = 64 lanes x 16 B, already in fragment order) is fetched by
    // wave i % 4 with global_load_lds (no VGPR round trip; LDS destination = uniform base + lane*16).
    auto issue_weights = [&](const uint4* src, int buf, int nks_) {
        char* dst = wlds + buf * wbuf_bytes;
        for (int i = wave; i < nks_ * NCO; i += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    // B-fragment base of this lane inside the LDS tile (pixel row 2*wave, column lx, upper half-wave = +16 B)
    const int bbase = ((wave * 2 * stride) * LW + lx * stride) * rec + hi * 16;
    const int brow = stride * LW * rec;      // second pixel row of this wave

    const int n_chunks = d->n_chunks;
    for (int c = 0; c < n_chunks; ++c) {
        const demfi_chunk& ch = d->chunks[c];
        if (c > 0) __syncthreads();          // all waves done reading the previous chunk's tile / weight ring
        const int nks = ch.nks;
        const int wtap_vecs = nks * NCO * 64;                       // 16-byte vectors of one tap's weights
        const uint4* __restrict__ wchunk = wbase + ch.w_off;
        issue_weights(wchunk, 0, nks);                              // overlaps the tile staging below
        // ---------------- stage the haloed input tile of this chunk into LDS ----------------------------
        for (int pi = ch.first_piece; pi < ch.first_piece + ch.n_pieces; ++pi) {
            const demfi_piece& p = d->pieces[pi];
            const char* src = (const char*)p.v.ptr;
            const int ush = p.up_shift;
            if (p.fat) {
                const int vpp = (p.nch * ESZ) >> 4;                 // 16-byte vectors per pixel: 1,2,4,8
                const int vsh = 31 - __builtin_clz(vpp);
                const int nitems = NP << vsh;
                const int64_t sx = p.v.sx * ESZ, sy = p.v.sy * ESZ;
                const char* srcb = src + (int64_t)bimg * p.v.sb * ESZ;
                const int ldsoff = p.lds_ch * ESZ;
                for (int it = tid; it < nitems; it += NT) {
                    const int px = it >> vsh;
                    const int v = it & (vpp - 1);
                    const int ly = __umulhi((uint32_t)px, lw_magic);
                    const int lxx = px - ly * LW;
                    const int iy = iy0 + ly, ix = ix0 + lxx;
                    uint4 val = make_uint4(0, 0, 0, 0);
                    if (src != nullptr && iy >= 0 && iy < inH && ix >= 0 && ix < inW)
                        val = *(const uint4*)(srcb + (iy >> ush) * sy + (ix >> ush) * sx + v * 16);
                    *(uint4*)(smem + px * rec + ldsoff + v * 16) = val;
                }
            } else {
                const int nch = p.nch;
                const bool f32src = p.v.is_f32 != 0;
                const int64_t sb = (int64_t)bimg * p.v.sb;
                for (int cc = 0; cc < nch; ++cc) {
                    const int64_t coff = sb + (int64_t)cc * p.v.sc;
                    const int ldsoff = (p.lds_ch + cc) * ESZ;
                    for (int px = tid; px < NP; px += NT) {
                        const int ly = __umulhi((uint32_t)px, lw_magic);
                        const int lxx = px - ly * LW;
                        const int iy = iy0 + ly, ix = ix0 + lxx;
                        float val = 0.0f;
                        if (src != nullptr && iy >= 0 && iy < inH && ix >= 0 && ix < inW) {
                            const int64_t off = coff + (int64_t)(iy >> ush) * p.v.sy + (int64_t)(ix >> ush) * p.v.sx;
                            val = f32src ? ((const float*)src)[off] : (float)((const half_t*)src)[off];
                        }
                        *(T*)(smem + px * rec + ldsoff) = (T)val;
                    }
                }
            }
        }
        __syncthreads();                         // tile staged, tap-0 weights landed (the barrier drains vmcnt)
        // ---------------- MFMA over taps x k-steps -------------------------------------------------------
        // A fragments come from the LDS weight ring (filled by LDS-DMA one tap ahead, shared by the 4 waves),
        // B fragments from the staged input tile.
        for (int tap = 0; tap < ntaps; ++tap) {
            if (tap + 1 < ntaps) issue_weights(wchunk + (int64_t)(tap + 1) * wtap_vecs, (tap + 1) & 1, nks);
            const char* wl = wlds + (tap & 1) * wbuf_bytes + lane * 16;
            const int ky = tap / kw, kx = tap - ky * kw;
            const int boff = bbase + (ky * LW + kx) * rec;
#pragma unroll 2
            for (int ks = 0; ks < nks; ++ks) {
                uint4 a[NCO];
#pragma unroll
                for (int s = 0; s < NCO; ++s) a[s] = *(const uint4*)(wl + (ks * NCO + s) * 1024);
                const uint4 b0 = *(const uint4*)(smem + boff + ks * 32);
                const uint4 b1 = *(const uint4*)(smem + boff + brow + ks * 32);
#pragma unroll
                for (int s = 0; s < NCO; ++s) {
                    Mma<T>::run(acc[s][0], a[s], b0);
                    Mma<T>::run(acc[s][1], a[s], b1);
                }
            }
            if (tap + 1 < ntaps) __syncthreads();     // next tap's weights landed; this tap's buffer is free
        }
    }

    // ---------------- epilogue --------------------------------------------------------------------------
    // acc[s][p][r]: pixel (oy0 + 2*wave + p, ox0 + lx), packed cout (cblk*NCO+s)*32 + 8*(r>>2) + 4*hi + (r&3)
    const float* __restrict__ bias = d->bias;
    const int ox = ox0 + lx;

    // ---- staged path: a 32-cout subtile whose 4 octets form one NHWC run of the path dtype goes through a
    // wave-private LDS transpose so that every lane owns 8 consecutive channels of one pixel: residual / gate
    // loads and the store are 16-byte (fp16) or 2x16-byte (fp32) accesses covering whole 64-byte runs per
    // pixel, instead of 8-byte accesses at a 128-byte lane stride.
    __syncthreads();                                   // every wave is done reading the input tile
    float* stage = (float*)(smem + wave * (64 * STAGE_LD * 4));
    static_for<0, NCO>([&](auto S) {
        constexpr int s = decltype(S)::value;
        const int sub = cblk * NCO + s;
        const int segi = d->sub_seg[sub];
        if (segi < 0) return;                          // uniform
        const demfi_seg& sg = d->segs[segi];
        const int ch0 = d->oct_ch[sub * 4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4_t bq = *(const f4_t*)(bias + sub * 32 + g * 8 + 4 * hi);
                f4_t v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[s][p][g * 4 + j] + bq[j];
                *(f4_t*)(stage + (p * 32 + lx) * STAGE_LD + g * 8 + 4 * hi) = v;
            }
        }
        __syncthreads();
        const int mode = sg.mode, act = sg.act;
        const T* resp = (const T*)sg.res.ptr;
        const T* auxp = (const T*)sg.aux.ptr;
        T* dstp = (T*)sg.dst.ptr;
        const int q = lane & 3;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int px = pass * 16 + (lane >> 2);
            const int oy = oy0 + wave * 2 + (px >> 5);
            const int oxx = ox0 + (px & 31);
            if (oy >= H || oxx >= W) continue;
            float v[8];
            {
                const f4_t v0 = *(const f4_t*)(stage + px * STAGE_LD + q * 8);
                const f4_t v1 = *(const f4_t*)(stage + px * STAGE_LD + q * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
            }
            const int cq = ch0 + q * 8;
            if (resp != nullptr) {
                float r[8];
                load8<T>(resp + (int64_t)bimg * sg.res.sb + (int64_t)oy * sg.res.sy + (int64_t)oxx * sg.res.sx + cq, r);
                if (mode == DEMFI_MODE_STORE) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j] + r[j], act);
                } else if (mode == DEMFI_MODE_MUL) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = sigmoidf_(v[j]) * r[j];
                } else {
                    float z[8];
                    load8<T>(auxp + (int64_t)bimg * sg.aux.sb + (int64_t)oy * sg.aux.sy + (int64_t)oxx * sg.aux.sx + cq, z);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (1.0f - z[j]) * r[j] + z[j] * tanhf(v[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], act);
            }
            const int dyy = oy * sg.scale + sg.dy, dxx = oxx * sg.scale + sg.dx;
            store8<T>(dstp + (int64_t)bimg * sg.dst.sb + (int64_t)dyy * sg.dst.sy + (int64_t)dxx * sg.dst.sx + cq, v);
        }
        __syncthreads();
    });

    // ---- direct path (thin / planar / ragged destinations): straight from the accumulator layout ----------
    static_for<0, NCO * 4>([&](auto SG) {
        {
            constexpr int s = decltype(SG)::value >> 2;
            constexpr int g = decltype(SG)::value & 3;
            if (d->sub_seg[cblk * NCO + s] >= 0) return;
            const int oct = (cblk * NCO + s) * 4 + g;
            const int on = d->oct_n[oct];
            if (on == 0) return;
            const demfi_seg& sg = d->segs[d->oct_seg[oct]];
            const int nq = min(max(on - 4 * hi, 0), 4);               // valid channels of this lane's quad
            const int cq = d->oct_ch[oct] + 4 * hi;                   // first channel inside the seg's views
            const f4_t bq = *(const f4_t*)(bias + oct * 8 + 4 * hi);
            const int mode = sg.mode, act = sg.act;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int oy = oy0 + wave * 2 + p;
                if (oy >= H || ox >= W || nq == 0) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[s][p][g * 4 + j] + bq[j];
                const bool hasres = sg.res.ptr != nullptr;
                if (hasres) {
                    const int64_t ro = (int64_t)bimg * sg.res.sb + (int64_t)oy * sg.res.sy + (int64_t)ox * sg.res.sx
                                       + (int64_t)cq * sg.res.sc;
                    float r[4];
                    if (sg.res.sc == 1 && nq == 4 && !sg.res.is_f32) {
                        h4_t rv = *(const h4_t*)((const half_t*)sg.res.ptr + ro);
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[j] = (float)rv[j];
                    } else if (sg.res.sc == 1 && nq == 4) {
                        f4_t rv = *(const f4_t*)((const float*)sg.res.ptr + ro);
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[j] = rv[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[j] = j < nq ? view_load(sg.res, ro + j * sg.res.sc) : 0.0f;
                    }
                    if (mode == DEMFI_MODE_STORE) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j] + r[j], act);
                    } else if (mode == DEMFI_MODE_MUL) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = sigmoidf_(v[j]) * r[j];
                    } else {   // GRU: (1-z)*h + z*tanh(v)
                        const int64_t ao = (int64_t)bimg * sg.aux.sb + (int64_t)oy * sg.aux.sy
                                           + (int64_t)ox * sg.aux.sx + (int64_t)cq * sg.aux.sc;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float z = j < nq ? view_load(sg.aux, ao + j * sg.aux.sc) : 0.0f;
                            v[j] = (1.0f - z) * r[j] + z * tanhf(v[j]);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], act);
                }
                const int dyy = oy * sg.scale + sg.dy, dxx = ox * sg.scale + sg.dx;
                const int64_t dofs = (int64_t)bimg * sg.dst.sb + (int64_t)dyy * sg.dst.sy + (int64_t)dxx * sg.dst.sx
                                     + (int64_t)cq * sg.dst.sc;
                if (sg.dst.sc == 1 && nq == 4 && !sg.dst.is_f32) {
                    h4_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
                    *(h4_t*)((half_t*)sg.dst.ptr + dofs) = o;
                } else if (sg.dst.sc == 1 && nq == 4) {
                    f4_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = v[j];
                    *(f4_t*)((float*)sg.dst.ptr + dofs) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < nq) view_store(sg.dst, dofs + j * sg.dst.sc, v[j]);
                }
            }
        }
    });
}

template <typename T, int NCO>
int launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, size_t lds)
{
    static bool attr_done = false;
    if (!attr_done) {
        DEMFI_HIP_CHECK(hipFuncSetAttribute((const void*)conv_kernel<T, NCO>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const int tiles = ((h->W + TW - 1) / TW) * ((h->H + TH - 1) / TH);
    dim3 grid(tiles, h->cout_pad / (32 * h->nco), h->batch);
    hipLaunchKernelGGL((conv_kernel<T, NCO>), grid, dim3(NT), lds, st, dev);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

template <typename T>
int dispatch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, size_t lds)
{
    switch (h->nco) {
    case 1: return launch<T, 1>(h, dev, st, lds);
    case 2: return launch<T, 2>(h, dev, st, lds);
    case 3: return launch<T, 3>(h, dev, st, lds);
    case 4: return launch<T, 4>(h, dev, st, lds);
    case 5: return launch<T, 5>(h, dev, st, lds);
    }
    return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: nco=%d not in 1..5", h->nco);
}

}  // namespace

extern "C" int64_t demfi_conv_lds_bytes(const demfi_conv* h)
{
    const int64_t LW = (int64_t)(TW - 1) * h->stride + h->kw;
    const int64_t LH = (int64_t)(TH - 1) * h->stride + h->kh;
    const int64_t tile = ((LW * LH * (h->rec_bytes + REC_PAD) + 1023) & ~1023ll)
                         + 2ll * (h->rec_bytes / 32) * h->nco * 1024;        // input tile + 2-tap weight ring
    const int64_t stage = 4 * 64 * STAGE_LD * 4;                 // 4 waves x 64 pixels x 36 floats
    return tile > stage ? tile : stage;
}

extern "C" int demfi_conv2d(const demfi_conv* h, const demfi_conv* dev, void* stream)
{
    if (!h || !dev) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: null descriptor");
    if (h->dtype != DEMFI_F16 && h->dtype != DEMFI_F32) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: dtype");
    if (h->stride != 1 && h->stride != 2) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: stride %d", h->stride);
    if (h->cout_pad <= 0 || h->cout_pad > 256 || h->cout_pad % (32 * h->nco))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: cout_pad=%d nco=%d", h->cout_pad, h->nco);
    if (h->rec_bytes != 32 && h->rec_bytes != 64 && h->rec_bytes != 128)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: rec_bytes=%d", h->rec_bytes);
    if (h->n_chunks < 1 || h->n_chunks > DEMFI_MAX_CHUNKS || h->n_pieces > DEMFI_MAX_PIECES || h->n_segs > DEMFI_MAX_SEGS)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: chunk/piece/seg count");
    const int64_t LW = (int64_t)(TW - 1) * h->stride + h->kw;
    const int64_t lds = demfi_conv_lds_bytes(h);
    if (lds > 160 * 1024) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: LDS tile %lld B > 160 KiB", (long long)lds);
    if (LW * ((int64_t)(TH - 1) * h->stride + h->kh) >= 65536) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: tile too large");
    if (h->lw_magic != (uint32_t)((0x100000000ull + LW - 1) / LW))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: lw_magic mismatch");
    const int esz = h->dtype == DEMFI_F16 ? 2 : 4;
    for (int c = 0; c < h->n_chunks; ++c) {
        const demfi_chunk& ch = h->chunks[c];
        if (ch.nks < 1 || ch.nks * 32 > h->rec_bytes) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: chunk %d nks", c);
        int used = 0;
        for (int pi = ch.first_piece; pi < ch.first_piece + ch.n_pieces; ++pi) {
            const demfi_piece& p = h->pieces[pi];
            if (p.lds_ch * esz != used) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: piece %d not contiguous in chunk", pi);
            if (p.fat) {
                const int bytes = p.nch * esz, vpp = bytes / 16;
                if (bytes % 16 || (vpp & (vpp - 1)) || vpp > 8 || p.v.sc != 1 || (p.v.is_f32 != (h->dtype == DEMFI_F32)))
                    return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: fat piece %d malformed", pi);
            }
            used += p.nch * esz;
        }
        if (used != ch.nks * 32) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: chunk %d covers %d B, expected %d", c, used, ch.nks * 32);
    }
    for (int sb = 0; sb < h->cout_pad / 32; ++sb) {
        const int sgi = h->sub_seg[sb];
        if (sgi < 0) continue;
        if (sgi >= h->n_segs) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: sub_seg[%d]=%d", sb, sgi);
        const demfi_seg& sg = h->segs[sgi];
        const int f32 = h->dtype == DEMFI_F32;
        bool ok = sg.dst.ptr && sg.dst.sc == 1 && sg.dst.is_f32 == f32;
        if (sg.res.ptr) ok = ok && sg.res.sc == 1 && sg.res.is_f32 == f32;
        if (sg.mode == DEMFI_MODE_GRU) ok = ok && sg.aux.ptr && sg.aux.sc == 1 && sg.aux.is_f32 == f32;
        if (sg.mode != DEMFI_MODE_STORE) ok = ok && sg.res.ptr;
        for (int o = 0; o < 4; ++o)
            ok = ok && h->oct_seg[sb * 4 + o] == sgi && h->oct_n[sb * 4 + o] == 8 &&
                 h->oct_ch[sb * 4 + o] == h->oct_ch[sb * 4] + 8 * o;
        ok = ok && h->oct_ch[sb * 4] % 8 == 0;
        if (!ok) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: subtile %d is not eligible for the staged epilogue", sb);
    }
    hipStream_t st = (hipStream_t)stream;
    return h->dtype == DEMFI_F16 ? dispatch<half_t>(h, dev, st, (size_t)lds) : dispatch<float>(h, dev, st, (size_t)lds);
}
