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
#include <stdlib.h>

namespace {

// Run-time experiment switches of the persistent kernels (env DEMFI_KNOB, read once on the host and copied into this
// word; 0 = product behaviour):  bit 0: DMA waves at s_setprio 3;  bit 1 (pair kernel): epilogue at priority 2, MFMA
// phase at 0;  bit 2 (pair kernel): MFMA phase at priority 2, epilogue at 0.
#if defined(DEMFI_ABLATION) || defined(DEMFI_TRACE)
__device__ int g_knob = 0;
#define DEMFI_KNOB_BIT(b) (g_knob & (b))
#else
#define DEMFI_KNOB_BIT(b) 0                                      // product build: no device global, no lazy hipMemcpyToSymbol in a launch path
#endif

// In-kernel phase trace (libdemfi_hip_trace.so, build.sh --trace; never in the product): s_memtime stamps of the first
// TR_TILES tiles of workgroups 0..TR_WGS-1, [wg][wave][tile][stamp].  MFMA waves: 0 = arrived at barrier A, 1 = released,
// 2 = MFMA phase done, 3 = epilogue issued.  DMA waves: 0 = tile landed (vmcnt 0), 1 = released, 2 = next tile issued.
#ifdef DEMFI_TRACE
constexpr int TR_WGS = 32, TR_WAVES = 10, TR_TILES = 24, TR_STAMPS = 6;
__device__ unsigned long long g_trace[TR_WGS * TR_WAVES * TR_TILES * TR_STAMPS];
#define TRACE_STAMP(wave_, k_, i_)                                                                                  \
    do {                                                                                                              \
        if (blockIdx.x < TR_WGS && (k_) < TR_TILES && (threadIdx.x & 63) == 0)                                      \
            g_trace[((blockIdx.x * TR_WAVES + (wave_)) * TR_TILES + (k_)) * TR_STAMPS + (i_)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define TRACE_STAMP(wave_, k_, i_) do { } while (0)
#endif

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
    // first MFMA of an accumulator with an explicit C operand (the bias rows: saves the epilogue's bias adds)
    static __device__ __forceinline__ void initc(f16x_t& acc, const uint4& a, const uint4& b, const f16x_t& c)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
    }
    // first MFMA of an accumulator: C = inline constant 0 instead of 16 v_mov per accumulator before the loop
    static __device__ __forceinline__ void init(f16x_t& acc, const uint4& a, const uint4& b)
    {
        const f16x_t z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), z, 0, 0, 0);
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
    const h8_t v = *gcp<h8_t>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float* o)
{
    const f4_t a = *gcp<f4_t>(p), b = *gcp<f4_t>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v);
template <> __device__ __forceinline__ void store8<half_t>(half_t* p, const float* v)
{
    h8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
    *gp<h8_t>(p) = o;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v)
{
    f4_t a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
    *gp<f4_t>(p) = a;
    *gp<f4_t>(p + 4) = b;
}

// Activation of N values behind ONE wave-uniform switch (a per-element switch compiles to a maze of scalar
// branches: ~8 s_cbranch per element dominated the epilogue).
template <int N>
__device__ __forceinline__ void apply_act_n(float (&v)[N], int act)
{
    switch (act) {
    case DEMFI_ACT_RELU:
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = fmaxf(v[j], 0.0f);
        break;
    case DEMFI_ACT_TANH:
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = fast_tanh(v[j]);
        break;
    case DEMFI_ACT_SIGMOID:
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = fast_sigmoid(v[j]);
        break;
    default: break;
    }
}

// ---- epilogue shared by the general and the persistent kernel -------------------------------------------
// acc[s][p][r]: pixel (oy0 + 2*wave + p, ox0 + lx), packed cout (cblk*NCO+s)*32 + 8*(r>>2) + 4*hi + (r&3).
// 'smem' must be free for reuse (the caller has synchronised the workgroup after the last tile read).
template <typename T, int NCO, bool BLOCK_SYNC, bool DIRECT = true>
__device__ __forceinline__ void conv_epilogue(const demfi_conv* __restrict__ d, f16x_t (&acc)[NCO][2], char* smem,
                                              int wave, int lane, int cblk, int bimg, int oy0, int ox0, int H, int W)
{
    const int hi = lane >> 5;
    const int lx = lane & 31;
    const float* __restrict__ bias = d->bias;
    const int ox = ox0 + lx;

    // ---- staged path: a 32-cout subtile whose 4 octets form one NHWC run of the path dtype goes through a
    // wave-private LDS transpose so that every lane owns 8 consecutive channels of one pixel: residual / gate
    // loads and the store are 16-byte (fp16) or 2x16-byte (fp32) accesses covering whole 64-byte runs per
    // pixel, instead of 8-byte accesses at a 128-byte lane stride.
    if constexpr (BLOCK_SYNC) __syncthreads();         // every wave is done reading the input tile
    // The staging area is wave-private: LDS instructions of one wave execute in order, so the write -> read ->
    // rewrite sequence below needs no workgroup barrier, only a compiler scheduling fence.
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
                const f4_t bq = *gcp<f4_t>(bias + sub * 32 + g * 8 + 4 * hi);
                f4_t v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[s][p][g * 4 + j] + bq[j];
                *(f4_t*)(stage + (p * 32 + lx) * STAGE_LD + g * 8 + 4 * hi) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
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
                    for (int j = 0; j < 8; ++j) v[j] = v[j] + r[j];
                    apply_act_n<8>(v, act);
                } else if (mode == DEMFI_MODE_MUL) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fast_sigmoid(v[j]) * r[j];
                } else {
                    float z[8];
                    load8<T>(auxp + (int64_t)bimg * sg.aux.sb + (int64_t)oy * sg.aux.sy + (int64_t)oxx * sg.aux.sx + cq, z);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (1.0f - z[j]) * r[j] + z[j] * fast_tanh(v[j]);
                }
            } else {
                apply_act_n<8>(v, act);
            }
            const int dyy = oy * sg.scale + sg.dy, dxx = oxx * sg.scale + sg.dx;
            store8<T>(dstp + (int64_t)bimg * sg.dst.sb + (int64_t)dyy * sg.dst.sy + (int64_t)dxx * sg.dst.sx + cq, v);
        }
        __builtin_amdgcn_wave_barrier();
    });

    // ---- direct path (thin / planar / ragged destinations): straight from the accumulator layout ----------
    // (DIRECT = false: every subtile of the layer takes the staged path -- the launch checks it -- and this generic per-octet code, two
    //  thirds of the kernel's 77-188 KB of instructions, is not instantiated: round 4, instruction-cache footprint)
    if constexpr (DIRECT)
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
            const f4_t bq = *gcp<f4_t>(bias + oct * 8 + 4 * hi);
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
                        h4_t rv = *gcp<h4_t>((const half_t*)sg.res.ptr + ro);
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[j] = (float)rv[j];
                    } else if (sg.res.sc == 1 && nq == 4) {
                        f4_t rv = *gcp<f4_t>((const float*)sg.res.ptr + ro);
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[j] = rv[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[j] = j < nq ? view_load(sg.res, ro + j * sg.res.sc) : 0.0f;
                    }
                    if (mode == DEMFI_MODE_STORE) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] + r[j];
                        apply_act_n<4>(v, act);
                    } else if (mode == DEMFI_MODE_MUL) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fast_sigmoid(v[j]) * r[j];
                    } else {   // GRU: (1-z)*h + z*tanh(v)
                        const int64_t ao = (int64_t)bimg * sg.aux.sb + (int64_t)oy * sg.aux.sy
                                           + (int64_t)ox * sg.aux.sx + (int64_t)cq * sg.aux.sc;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float z = j < nq ? view_load(sg.aux, ao + j * sg.aux.sc) : 0.0f;
                            v[j] = (1.0f - z) * r[j] + z * fast_tanh(v[j]);
                        }
                    }
                } else {
                    apply_act_n<4>(v, act);
                }
                const int dyy = oy * sg.scale + sg.dy, dxx = ox * sg.scale + sg.dx;
                const int64_t dofs = (int64_t)bimg * sg.dst.sb + (int64_t)dyy * sg.dst.sy + (int64_t)dxx * sg.dst.sx
                                     + (int64_t)cq * sg.dst.sc;
                if (sg.dst.sc == 1 && nq == 4 && !sg.dst.is_f32) {
                    h4_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
                    *gp<h4_t>((half_t*)sg.dst.ptr + dofs) = o;
                } else if (sg.dst.sc == 1 && nq == 4) {
                    f4_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = v[j];
                    *gp<f4_t>((float*)sg.dst.ptr + dofs) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < nq) view_store(sg.dst, dofs + j * sg.dst.sc, v[j]);
                }
            }
        }
    });
}

// second launch_bounds argument = minimum waves per SIMD: 2-3 resident workgroups per CU let one workgroup's
// tile staging overlap another's MFMA phase.
template <typename T, int NCO, bool DIRECT = true>
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
#ifdef DEMFI_ABLATION
            if (DEMFI_KNOB_BIT(32)) break;                              // experiment: no tile staging (garbage operands): what the staging costs
#endif
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
                // STG_UNR independent loads in flight per thread before the first LDS write (round 3: the one-item loop serialised a
                // global-memory round trip per item -- 10 per thread for a 3x3 tile of 128-byte records -- and cost 25-70 % of the
                // general kernel's layers: profiles/r03_notes.md section 13)
                constexpr int STG_UNR = 4;
                // interior tiles of tensors below 4 GiB per image (every tile but the frame's border): no bounds tests, 32-bit offsets
                // from a uniform base (the saddr form of the load: no 64-bit address arithmetic per item)
                const bool fast = src != nullptr && iy0 >= 0 && iy0 + LH <= inH && ix0 >= 0 && ix0 + LW <= inW &&
                                  (uint64_t)(((inH - 1) >> ush) + 1) * (uint64_t)sy < ((uint64_t)1 << 32) && sy >= 0 && sx >= 0;
                if (fast) {
                    const uint32_t sy32 = (uint32_t)sy, sx32 = (uint32_t)sx;
                    for (int it0 = tid; it0 < nitems; it0 += STG_UNR * NT) {
                        uint4 val[STG_UNR];
                        int dsto[STG_UNR];
#pragma unroll
                        for (int u = 0; u < STG_UNR; ++u) {
                            const int it = min(it0 + u * NT, nitems - 1);       // clamped: an unconditional load (re-reads the last item)
                            const int px = it >> vsh;
                            const int v = it & (vpp - 1);
                            const int ly = __umulhi((uint32_t)px, lw_magic);
                            const int lxx = px - ly * LW;
                            dsto[u] = it0 + u * NT < nitems ? px * rec + ldsoff + v * 16 : -1;
                            val[u] = ld_global16(srcb + (uint32_t)(((uint32_t)(iy0 + ly) >> ush) * sy32 + ((uint32_t)(ix0 + lxx) >> ush) * sx32 + v * 16));
                        }
#pragma unroll
                        for (int u = 0; u < STG_UNR; ++u)
                            if (dsto[u] >= 0) *(uint4*)(smem + dsto[u]) = val[u];
                    }
                } else
                for (int it0 = tid; it0 < nitems; it0 += STG_UNR * NT) {
                    uint4 val[STG_UNR];
                    int dsto[STG_UNR];
#pragma unroll
                    for (int u = 0; u < STG_UNR; ++u) {
                        const int it = it0 + u * NT;
                        const int px = it >> vsh;
                        const int v = it & (vpp - 1);
                        const int ly = __umulhi((uint32_t)px, lw_magic);
                        const int lxx = px - ly * LW;
                        const int iy = iy0 + ly, ix = ix0 + lxx;
                        val[u] = make_uint4(0, 0, 0, 0);
                        dsto[u] = it < nitems ? px * rec + ldsoff + v * 16 : -1;
                        if (it < nitems && src != nullptr && iy >= 0 && iy < inH && ix >= 0 && ix < inW)
                            val[u] = ld_global16(srcb + (iy >> ush) * sy + (ix >> ush) * sx + v * 16);
                    }
#pragma unroll
                    for (int u = 0; u < STG_UNR; ++u)
                        if (dsto[u] >= 0) *(uint4*)(smem + dsto[u]) = val[u];
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
                            val = f32src ? gcp<float>(src)[off] : (float)gcp<half_t>(src)[off];
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
    conv_epilogue<T, NCO, true, DIRECT>(d, acc, smem, wave, lane, cblk, bimg, oy0, ox0, H, W);
}


// ======================================================================================================
// Persistent specialisation for the workhorse shape of the network: fp16, stride 1, KHxKW filter, ONE NHWC
// input of 64 channels (128-byte pixel records), <= 64 output channels (all 3x3 64->64 layers of the FAC-FB
// encoder, D1 and D2: ~52 % of the MACs of a forward).
//   * one workgroup per CU walks many 8x32 output tiles (XCD-aware bands);
//   * ALL filter taps stay resident in LDS for the whole launch (72 KiB for 3x3x64x64) -> no per-tap weight
//     traffic and no per-tap barriers;
//   * the haloed input tile is fetched by LDS-DMA (global_load_lds, no VGPR round trip, zero padding through a
//     zero page) into a double buffer: tile k+1 streams in while tile k is on the matrix cores;
//   * the epilogue needs no LDS and no cross-lane traffic: the layers of this kernel are packed in a permuted cout order
//     (demfi_conv.cout_perm) in which the two accumulator quads a lane owns are 8 consecutive output channels of its
//     pixel (16-byte stores / residual loads straight from registers), so there is ONE barrier per tile;
//   * pixel records are unpadded (128 B); bank conflicts are removed by an XOR swizzle of the 16-byte slot,
//     applied on the DMA's per-lane SOURCE address and on the ds_read address (the LDS image stays lane-linear).
// ======================================================================================================
constexpr int P_LW = TW + 2, P_LH = TH + 2;                    // 3x3 halo
constexpr int P_NP = P_LW * P_LH;                               // 340 pixels
constexpr int P_NI = (P_NP + 7) / 8;                            // 43 DMA instructions (8 pixels x 8 slots each)
constexpr int P_TILE_BYTES = P_NI * 1024;                       // 44,032 B per buffer

constexpr int P_NT = NT + 64;                                   // 4 MFMA waves + 1 DMA wave

// One k-step pair of fragments: 2 k-steps x (NCO A fragments + 2 B fragments)
template <int NCO> struct FragSet { uint4 a[2][NCO]; uint4 b[2][2]; };

#ifndef DEMFI_P_NDMA
#define DEMFI_P_NDMA 2
#endif
#ifndef DEMFI_P_KYREUSE
#define DEMFI_P_KYREUSE 1        // 0: one (tap, k-step pair) at a time, 12 ds_reads per 12 MFMAs (A/B builds)
#endif
constexpr int P_NDMA = DEMFI_P_NDMA;                            // waves issuing the tile DMA (instruction i -> wave i % P_NDMA)
template <int NCO, int VAR, bool RES = true>   // RES: the segment has a residual input (compile time: keeps the loads free of phis).  VAR: 0 = product; 1 no epilogue, 2 no MFMA phase, 3 no tile DMA, 4 epilogue only (ablation builds)
__global__ __launch_bounds__(NT + 64 * P_NDMA, 1) void conv3x3_c64_persist_kernel(const demfi_conv* __restrict__ d)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTAPS = 9, NKS = 4;
    constexpr int WBYTES = NTAPS * NKS * NCO * 1024;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d->H, W = d->W;
    const int tiles_x = (W + TW - 1) / TW;
    const int tiles_y = (H + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int total = tiles_img * d->batch;
    char* const wlds = smem;                                    // resident weights
    char* const tbuf = smem + WBYTES;                           // 2 x tile buffer

    // tile sequence of this workgroup: XCD x = b & 7 owns the contiguous band [lo, hi) of tile indices
    const int G = gridDim.x;
    int t_first, t_end, t_step;
    if ((G & 7) == 0 && total >= G) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        const int lo = xcd * q + min(xcd, r);
        t_first = lo + idx;
        t_end = lo + q + (xcd < r ? 1 : 0);
        t_step = G >> 3;
    } else {
        t_first = blockIdx.x;
        t_end = total;
        t_step = G;
    }
    if (t_first >= t_end) return;                               // uniform per workgroup

    auto tile_coords = [&](int t, int& bimg, int& oy0, int& ox0) {
        bimg = t / tiles_img;
        const int rem = t - bimg * tiles_img;
        const int ty = rem / tiles_x;
        oy0 = ty * TH;
        ox0 = (rem - ty * tiles_x) * TW;
    };

    if (wave >= 4) {
        // ================= DMA waves: own every global->LDS transfer, so only THEIR vmcnt tracks them ============
        if (DEMFI_KNOB_BIT(1)) __builtin_amdgcn_s_setprio(3);
        const int dw = wave - 4;
        const demfi_piece& pc = d->pieces[0];
        const char* const src = (const char*)pc.v.ptr;
        const int64_t sx = pc.v.sx * 2, sy = pc.v.sy * 2, sb = pc.v.sb * 2;
        const char* const zeros = (const char*)d->zero_page;
        // instruction i covers pixels 8i..8i+7; lane -> (pixel 8i + lane/8, physical 16-byte slot lane%8).
        // The per-lane byte offsets relative to the tile origin and the (row, column) pairs never change: compute
        // them once (86 VGPRs) so that issuing a tile is ~4 VALU per DMA instruction instead of ~50.
        int off[P_NI], lyx[P_NI];
#pragma unroll
        for (int i = 0; i < P_NI; ++i) {
            const int px = i * 8 + (lane >> 3);
            const int ly = px / P_LW;
            const int lxx = px - ly * P_LW;
            const int v = (lane & 7) ^ ((lxx >> 1) & 7);                // logical slot at this physical slot: swizzle by tile COLUMN
            off[i] = (int)(ly * sy + lxx * sx) + v * 16;
            lyx[i] = px < P_NP ? (ly | (lxx << 8)) : 0xffff;
        }
        auto issue_tile = [&](int t, int buf) {
            int bimg, oy0, ox0;
            tile_coords(t, bimg, oy0, ox0);
            const char* base = src + (int64_t)bimg * sb + (int64_t)(oy0 - 1) * sy + (int64_t)(ox0 - 1) * sx;
            char* dst = tbuf + buf * P_TILE_BYTES;
            const bool interior = oy0 >= 1 && oy0 + TH + 1 <= H && ox0 >= 1 && ox0 + TW + 1 <= W;
            if (interior) {
#pragma unroll
                for (int i = 0; i < P_NI; ++i) {
                    if ((i % P_NDMA) != dw) continue;            // wave-uniform
                    const char* g = (i == P_NI - 1 && lyx[i] == 0xffff) ? zeros : base + off[i];
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < P_NI; ++i) {
                    if ((i % P_NDMA) != dw) continue;
                    const int iy = oy0 - 1 + (lyx[i] & 255), ix = ox0 - 1 + (lyx[i] >> 8);
                    const char* g = (lyx[i] != 0xffff && iy >= 0 && iy < H && ix >= 0 && ix < W) ? base + off[i] : zeros;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                }
            }
        };
        const uint4* wsrc = (const uint4*)d->wpack;
        for (int i = dw; i < NTAPS * NKS * NCO; i += P_NDMA)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + i * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(wlds + i * 1024), 16, 0, 0);
        issue_tile(t_first, 0);
        int buf = 0;
        [[maybe_unused]] int trk = 0;
        for (int t = t_first; t < t_end; t += t_step, buf ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile t (and the weights) have landed in LDS
            TRACE_STAMP(wave, trk, 0);
            __syncthreads();                                    // A: hand tile t to the MFMA waves
            TRACE_STAMP(wave, trk, 1);
            if (VAR != 3 && VAR != 4 && VAR != 10 && t + t_step < t_end) issue_tile(t + t_step, buf ^ 1);   // streams in under the MFMAs
            TRACE_STAMP(wave, trk, 2);
            ++trk;
        }
        return;
    }

    // ================= MFMA waves ============================================================================
    const int hi = lane >> 5;
    const int lx = lane & 31;
    // ---- everything the epilogue needs from the descriptor, hoisted out of the tile loop (barriers are memory
    // fences: descriptor fields read inside the loop would be re-fetched through dependent scalar loads per tile)
    const demfi_seg& sg0 = d->segs[d->sub_seg[0]];
    half_t* const dstp = (half_t*)sg0.dst.ptr;
    const half_t* const resp = (const half_t*)sg0.res.ptr;
    const int64_t d_sx = sg0.dst.sx, d_sy = sg0.dst.sy, d_sb = sg0.dst.sb;
    const int64_t r_sx = sg0.res.sx, r_sy = sg0.res.sy, r_sb = sg0.res.sb;
    const float act_floor = sg0.act == DEMFI_ACT_RELU ? 0.0f : -__builtin_huge_valf();
    h8_t act_floor8;
#pragma unroll
    for (int j = 0; j < 8; ++j) act_floor8[j] = (half_t)act_floor;
    const int ch0 = d->oct_ch[0];
    // Epilogue layout (cout_perm): quads 2m and 2m+1 of lane (lx, hi) are the 8 consecutive output channels
    // s*32 + 16m + 8hi .. of pixel lx: 16-byte stores / residual loads straight from registers, no LDS transpose, no
    // lane exchange.  The bias (NCO*32 floats, MFMA-row order) sits in the 2 KiB of LDS behind the tile buffers.
    float* const bias_lds = (float*)(tbuf + 2 * P_TILE_BYTES);
    if (tid < NCO * 32) bias_lds[tid] = d->bias[tid];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the write is in LDS before this wave's first (raw) barrier A
    int boff[12];                                               // [kx*4 + ks]: (column lx+kx) record + swizzled 16-byte slot
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int col = lx + (g >> 2);
        boff[g] = col * 128 + ((((g & 3) * 2 + hi) ^ ((col >> 1) & 7)) << 4);
    }
    const char* const wl = wlds + lane * 16;
    int buf = 0;
    [[maybe_unused]] int trk = -1;
    for (int t = t_first; t < t_end; t += t_step, buf ^= 1) {
        int bimg, oy0, ox0;
        tile_coords(t, bimg, oy0, ox0);
        ++trk;
        // Residual of this tile: issued before the MFMA phase, consumed in the epilogue.  The loads are unconditional
        // (clamped address, no per-lane branch) and barrier A is a RAW s_barrier: a lane-divergent load leaves register
        // copies behind and __syncthreads() carries a fence -- either one makes the compiler put s_waitcnt vmcnt(0)
        // in front of the MFMA phase, i.e. a full HBM round trip per tile (measured: +0.064 ms on the 0.28 ms launch).
        u4_t rreg[NCO][2][2];
        if constexpr (RES) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int oy = min(oy0 + wave * 2 + p, H - 1), oxx = min(ox0 + lx, W - 1);
                const half_t* rp = resp + bimg * r_sb + oy * r_sy + oxx * r_sx + ch0 + hi * 8;
#pragma unroll
                for (int s = 0; s < NCO; ++s) {
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) rreg[s][p][m2] = *gcp<u4_t>(rp + s * 32 + m2 * 16);
                }
            }
        }
        // A: tile t is in LDS (the DMA wave waited for it).  These waves wrote no LDS and consumed every ds_read of the
        // previous tile, so no counter has to drain here; "memory" keeps the compiler from moving LDS reads above it.
        TRACE_STAMP(wave, trk, 0);
        asm volatile("s_barrier" ::: "memory");
        TRACE_STAMP(wave, trk, 1);
        f16x_t acc[NCO][2];
#pragma unroll
        for (int s = 0; s < NCO; ++s) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[s][0][i] = 0.0f; acc[s][1][i] = 0.0f; }
        }
        const char* tb = tbuf + buf * P_TILE_BYTES + (wave * 2) * (P_LW * 128);
        if (VAR != 2 && VAR != 4) {
            // software pipeline over 18 k-step pairs: the fragments of pair i+1 are in flight while the 4*NCO MFMAs
            // of pair i run (one wave per SIMD: nothing else hides the LDS latency)
            auto load_pair = [&](FragSet<NCO>& f, int pair) {
                const int tap = pair >> 1, ks0 = (pair & 1) * 2;
                const int ky = tap / 3, kx = tap % 3;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int ks = ks0 + k;
#pragma unroll
                    for (int s = 0; s < NCO; ++s) f.a[k][s] = *(const uint4*)(wl + ((tap * NKS + ks) * NCO + s) * 1024);
                    const char* p0 = tb + boff[kx * 4 + ks];            // row index is an immediate of the ds_read
                    f.b[k][0] = *(const uint4*)(p0 + ky * (P_LW * 128));
                    f.b[k][1] = *(const uint4*)(p0 + (ky + 1) * (P_LW * 128));
                }
            };
            auto mma_pair = [&](const FragSet<NCO>& f) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
#pragma unroll
                    for (int s = 0; s < NCO; ++s) {
                        Mma<half_t>::run(acc[s][0], f.a[k][s], f.b[k][0]);
                        Mma<half_t>::run(acc[s][1], f.a[k][s], f.b[k][1]);
                    }
                }
            };
            if constexpr (VAR == 0 && DEMFI_P_KYREUSE != 0) {
                // Input-row reuse across ky: for one (kx, k-step) the taps ky = 0..2 of output rows p = 0, 1 read input rows
                // p + ky = 0..3 at the same column offset -- 4 distinct B fragments feed 6 (ky, p) combinations.  One group =
                // 4 row fragments + 3*NCO weight fragments -> 6*NCO MFMAs: 10 ds_reads per 12 MFMAs instead of 12 (NCO = 2).
                struct RowFrag { uint4 a[3][NCO]; uint4 b[4]; };
                auto load_g = [&](RowFrag& f, int g) {          // g = kx*4 + ks
                    const int kx = g >> 2, ks = g & 3;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                        for (int s = 0; s < NCO; ++s) f.a[ky][s] = *(const uint4*)(wl + (((ky * 3 + kx) * NKS + ks) * NCO + s) * 1024);
                    }
                    const char* p0 = tb + boff[kx * 4 + ks];    // row index is an immediate of the ds_read
#pragma unroll
                    for (int r = 0; r < 4; ++r) f.b[r] = *(const uint4*)(p0 + r * (P_LW * 128));
                };
                auto mma_g = [&](const RowFrag& f) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                        for (int s = 0; s < NCO; ++s) {
                            Mma<half_t>::run(acc[s][0], f.a[ky][s], f.b[ky]);
                            Mma<half_t>::run(acc[s][1], f.a[ky][s], f.b[ky + 1]);
                        }
                    }
                };
                auto groups = [&](bool loads) {
#pragma unroll
                    for (int q = 0; q < 6 * NCO; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // 1 MFMA
                        if (loads && q < 3 * NCO + 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read of the next group
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                RowFrag f0, f1;
                load_g(f0, 0);
                static_for<0, 6>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    load_g(f1, 2 * i + 1);
                    mma_g(f0);
                    groups(true);
                    if constexpr (i < 5) load_g(f0, 2 * i + 2);
                    mma_g(f1);
                    groups(i < 5);
                });
            } else
            if constexpr (VAR == 11) {
                // ablation: ring of four k-step fragment sets, loads three k-steps (12 MFMAs) ahead of their use, one ds_read
                // issued per MFMA
                struct StepFrag { uint4 a[NCO]; uint4 b[2]; };
                auto load_step = [&](StepFrag& f, int g) {      // g = tap*4 + ks
                    const int tap = g >> 2, ks = g & 3;
                    const int ky = tap / 3, kx = tap % 3;
#pragma unroll
                    for (int s = 0; s < NCO; ++s) f.a[s] = *(const uint4*)(wl + (g * NCO + s) * 1024);
                    const char* p0 = tb + boff[kx * 4 + ks];
                    f.b[0] = *(const uint4*)(p0 + ky * (P_LW * 128));
                    f.b[1] = *(const uint4*)(p0 + (ky + 1) * (P_LW * 128));
                };
                StepFrag fr[4];
                load_step(fr[0], 0);
                load_step(fr[1], 1);
                load_step(fr[2], 2);
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 36>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    if constexpr (g + 3 < 36) load_step(fr[(g + 3) & 3], g + 3);
#pragma unroll
                    for (int s = 0; s < NCO; ++s) {
                        Mma<half_t>::run(acc[s][0], fr[g & 3].a[s], fr[g & 3].b[0]);
                        Mma<half_t>::run(acc[s][1], fr[g & 3].a[s], fr[g & 3].b[1]);
                    }
                    if constexpr (g + 3 < 36) {
#pragma unroll
                        for (int q = 0; q < 2 * NCO; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
            FragSet<NCO> f0, f1;
            load_pair(f0, 0);
            if constexpr (VAR == 7) load_pair(f1, 1);           // ablation: fragments loaded once per tile, no LDS traffic below
            static_for<0, 9>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (VAR != 7 && VAR != 9) {
                    // ds_reads of the next pair interleaved 1:1 with the MFMAs of this pair (sched_group_barrier): the matrix
                    // pipe does not idle while 8 ds_reads issue back to back (+3 % over the block schedule, VAR 9)
                    load_pair(f1, 2 * i + 1);
                    mma_pair(f0);
#pragma unroll
                    for (int q = 0; q < 4 * NCO; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (i < 8) load_pair(f0, 2 * i + 2);
                    mma_pair(f1);
#pragma unroll
                    for (int q = 0; q < 4 * NCO; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    return;
                }
                if constexpr (VAR != 7) load_pair(f1, 2 * i + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_pair(f0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i < 8 && VAR != 7) load_pair(f0, 2 * i + 2);
                __builtin_amdgcn_sched_barrier(0);
                mma_pair(f1);
                __builtin_amdgcn_sched_barrier(0);
            });
            }
        }
        // no second barrier: the epilogue works from registers, and tile t's buffer is only overwritten by the DMA of
        // tile t+2, issued after barrier A of tile t+1, which every MFMA wave reaches after this MFMA phase
        if (VAR == 1 || VAR == 10) {                            // 10: MFMA phase only (no tile DMA, no epilogue)
#pragma unroll
            for (int s = 0; s < NCO; ++s) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::"v"(acc[s][0]));
                asm volatile("" ::"v"(acc[s][1]));
#endif
            }
            continue;
        }
        // ---- epilogue straight from the accumulators (the tile buffer is not reused: barrier B only orders the DMA) ----
#if defined(DEMFI_TRACE) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int s = 0; s < NCO; ++s) { asm volatile("" ::"v"(acc[s][0])); asm volatile("" ::"v"(acc[s][1])); }
        TRACE_STAMP(wave, trk, 2);
#endif
        if constexpr (RES) {
            // Retire the residual loads HERE (they landed during the MFMA phase): otherwise the compiler's in-order
            // vmcnt bookkeeping makes the later units wait for this epilogue's own stores to be acknowledged.
#pragma unroll
            for (int s = 0; s < NCO; ++s) {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(rreg[s][q >> 1][q & 1]));
            }
        }
#pragma unroll
        for (int s = 0; s < NCO; ++s) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                // cout_perm: MFMA row (quad g, half hi, j) holds channel (g>>1)*16 + hi*8 + (g&1)*4 + j, so quads 2*m2 and 2*m2+1
                // of this lane are the 8 consecutive channels 16*m2 + 8*hi .. +7 of its pixel -- no cross-lane exchange (round 1
                // used a v_permlane32_swap per accumulator pair here); bias_lds is in MFMA-row order
                const f4_t b0 = *(const f4_t*)(bias_lds + s * 32 + (2 * m2) * 8 + hi * 4);
                const f4_t b1 = *(const f4_t*)(bias_lds + s * 32 + (2 * m2 + 1) * 8 + hi * 4);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[s][p][(2 * m2) * 4 + j] + b0[j];
                        v[4 + j] = acc[s][p][(2 * m2 + 1) * 4 + j] + b1[j];
                    }
                    if constexpr (RES) {
                        const h8_t r = __builtin_bit_cast(h8_t, rreg[s][p][m2]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += (float)r[j];
                    }
                    h8_t o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
                    o = __builtin_elementwise_max(o, act_floor8);   // ReLU or identity (floor -inf), branch-free; rounding is monotonic: max after the conversion gives the same value
                    const int oy = oy0 + wave * 2 + p, oxx = ox0 + lx;
                    if (oy < H && oxx < W)
                        *gp<u4_t>(dstp + bimg * d_sb + oy * d_sy + oxx * d_sx + ch0 + s * 32 + m2 * 16 + hi * 8) = __builtin_bit_cast(u4_t, o);
                }
            }
        }
        TRACE_STAMP(wave, trk, 3);
    }
}

template <int NCO, int VAR = 0>
int launch_persist(const demfi_conv* h, const demfi_conv* dev, hipStream_t st)
{
    const size_t lds = 9 * 4 * NCO * 1024 + 2 * P_TILE_BYTES + 1024;      // weights + 2 tiles + bias
    DEMFI_LDS_ATTR((conv3x3_c64_persist_kernel<NCO, VAR, true>));
    DEMFI_LDS_ATTR((conv3x3_c64_persist_kernel<NCO, VAR, false>));
    const int total = ((h->W + TW - 1) / TW) * ((h->H + TH - 1) / TH) * h->batch;
    const int grid = total >= 256 ? 256 : total;
    if (h->segs[h->sub_seg[0]].res.ptr != nullptr)
        hipLaunchKernelGGL((conv3x3_c64_persist_kernel<NCO, VAR, true>), dim3(grid), dim3(NT + 64 * P_NDMA), lds, st, dev);
    else
        hipLaunchKernelGGL((conv3x3_c64_persist_kernel<NCO, VAR, false>), dim3(grid), dim3(NT + 64 * P_NDMA), lds, st, dev);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}


// ======================================================================================================
// STAGED-STORE variant of the 64 -> 64 kernel (NCO == 2).
// What the in-kernel phase trace says about the 4-wave kernel above (profiles/r03_phase_trace.md; cycles at the ~1.7 GHz the part
// sustains under this load): a tile period of 7 700 cycles = MFMA phase 5 200 (144 MFMAs = 4 608 pipe cycles) + epilogue 2 200 +
// barrier ~300, and the epilogue is the CU's store path: 32 KiB per tile at ~16 B/clk/CU (tools/microbench/store_path_per_cu.hip)
// = 2 048 cycles during which the matrix pipe of all four SIMDs idles.  tools/microbench/overlap_matrix.hip says which waves may
// share a SIMD: an OLDER k-loop-like MFMA wave is not slowed by a YOUNGER wave that issues global stores (687 vs 683 cycles per
// 16 MFMAs, stores at full rate), while an older storing wave starves a younger MFMA wave completely -- so the roles are fixed by
// age: MFMA waves 0-3 never touch global memory for their outputs; after the MFMA phase they apply bias / residual / ReLU in
// registers, ds_write the packed fp16 tile into the tile buffer they have just finished reading (barrier B) and go on to the next
// tile.  The four helper waves (4-7, one per SIMD, younger) read the staged tile back 8 lanes per pixel, issue the global stores
// as whole 128-byte lines while the next tile is on the matrix cores, and then issue the LDS-DMA of tile k+2 into the same buffer
// (helper w drains exactly the 1-KiB chunks its own DMA instructions overwrite, so nothing else has to be synchronised).
// Round 2 built this once on the 2-DMA-wave kernel and measured nothing (profiles/r02_notes.md); the trace shows why: there the
// helper path (stores, then 3 300 cycles of DMA issue starved by the MFMA waves, then the landing) was as long as the period.
// ======================================================================================================
// (fp16 half of a packed pair) * 1.0 + c in one VALU op: the residual add of the epilogue
__device__ __forceinline__ float res_mix_lo(unsigned a, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}
__device__ __forceinline__ float res_mix_hi(unsigned a, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}
// c - (fp16 half of a packed pair) in one VALU op (no v_cvt_f32_f16): the GRU update's tanh(.) - h
__device__ __forceinline__ float sub_mix_lo(float c, unsigned a)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}
__device__ __forceinline__ float sub_mix_hi(float c, unsigned a)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}
// Streaming (nt) hints of the staged-store kernel.  Bit 1 (default): the helper waves' output stores -- whole 128-byte lines of tensors
// of hundreds of MB that the next launch re-reads from HBM anyway; without the hint the written lines compete with the input tiles for
// the L2s and the Infinity Cache: residual launches -1.5..-2 %, the window -0.6 ms (profiles/r04_notes.md section 11).  Experiment
// bits, both measured negative there: 2 = nt residual loads (+15 % on the residual launches), 4 = nt tile DMA.  The same hint on the
// 16-byte-per-lane stores of the MFMA waves of the GRU / narrow / streamed-weight kernels is 1.8x / 1.1x / 1.02x SLOWER (partial lines).
#ifndef DEMFI_STG_NT
#define DEMFI_STG_NT 1
#endif
#ifndef DEMFI_STG_RES_AHEAD
#define DEMFI_STG_RES_AHEAD 0                                    // 1: the residual of tile k+1 is fetched during tile k (two register sets: measured no better than 0 with an early issue point)
#endif
#ifndef DEMFI_STG_RES_AT
#define DEMFI_STG_RES_AT -1                                      // k-loop third after which the residual loads are issued (-1: before barrier A, at the head of the tile)
#endif
constexpr int SG_NH = 4;                                         // helper waves
constexpr int SG_NT = NT + 64 * SG_NH;
template <bool RES, bool TANH = false>   // TANH: tanh after the residual add (Refine_Module.dec3's feature halves, DeMFInet.py:86-87) instead of ReLU / identity
__global__ __launch_bounds__(SG_NT, 1) void conv3x3_c64_stg_kernel(const demfi_conv* __restrict__ d)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTAPS = 9, NKS = 4, NCO = 2;
    constexpr int WBYTES = NTAPS * NKS * NCO * 1024;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d->H, W = d->W;
    const int tiles_x = (W + TW - 1) / TW;
    const int tiles_y = (H + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int total = tiles_img * d->batch;
    char* const wlds = smem;
    char* const tbuf = smem + WBYTES;
    const int G = gridDim.x;
    int t_first, t_end, t_step;
    if ((G & 7) == 0 && total >= G) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        const int lo = xcd * q + min(xcd, r);
        t_first = lo + idx;
        t_end = lo + q + (xcd < r ? 1 : 0);
        t_step = G >> 3;
    } else {
        t_first = blockIdx.x;
        t_end = total;
        t_step = G;
    }
    if (t_first >= t_end) return;
    auto tile_coords = [&](int t, int& bimg, int& oy0, int& ox0) {
        bimg = t / tiles_img;
        const int rem = t - bimg * tiles_img;
        const int ty = rem / tiles_x;
        oy0 = ty * TH;
        ox0 = (rem - ty * tiles_x) * TW;
    };
    const demfi_seg& sg0 = d->segs[d->sub_seg[0]];
    const int64_t d_sx = sg0.dst.sx, d_sy = sg0.dst.sy, d_sb = sg0.dst.sb;
    [[maybe_unused]] int trk = 0;

    if (wave >= 4) {
        // ================= helper waves: tile DMA + the global stores of the staged outputs ==========================
        if (DEMFI_KNOB_BIT(1)) __builtin_amdgcn_s_setprio(2);
        const int dw = wave - 4;
        constexpr int NIW = (P_NI + SG_NH - 1) / SG_NH;          // DMA instructions per helper (11; the last one may not exist)
        const demfi_piece& pc = d->pieces[0];
        const char* const src = (const char*)pc.v.ptr;
        const int64_t sx = pc.v.sx * 2, sy = pc.v.sy * 2, sb = pc.v.sb * 2;
        const char* const zeros = (const char*)d->zero_page;
        unsigned off[NIW];                                       // unsigned: uniform base + zero-extended 32-bit lane offset = the saddr form (no VALU per instruction)
        int lyx[NIW];
#pragma unroll
        for (int k = 0; k < NIW; ++k) {
            const int i = dw + SG_NH * k;
            const int px = i * 8 + (lane >> 3);
            const int pxc = min(px, P_NP - 1);                   // lanes past the tile (last instruction only) re-read its last pixel: never consumed
            const int ly = pxc / P_LW;
            const int lxx = pxc - ly * P_LW;
            const int v = (lane & 7) ^ ((lxx >> 1) & 7);
            off[k] = (unsigned)((ly + 1) * sy + (lxx + 1) * sx) + v * 16;      // relative to pixel (-2,-2) of the tile: never negative
            lyx[k] = (i < P_NI && px < P_NP) ? (ly | (lxx << 8)) : 0xffff;
        }
        auto issue_tile = [&](int t, int buf) {
            int bimg, oy0, ox0;
            tile_coords(t, bimg, oy0, ox0);
            const char* base = src + (int64_t)bimg * sb + (int64_t)(oy0 - 2) * sy + (int64_t)(ox0 - 2) * sx;
            char* dst = tbuf + buf * P_TILE_BYTES;
            const bool interior = oy0 >= 1 && oy0 + TH + 1 <= H && ox0 >= 1 && ox0 + TW + 1 <= W;
            if (interior) {                                      // uniform base + precomputed 32-bit lane offset: ~2 VALU per instruction
#pragma unroll
                for (int k = 0; k < NIW; ++k) {
                    const int i = dw + SG_NH * k;
                    if (i >= P_NI) continue;                     // wave-uniform
                    const char* g = base + off[k];
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, (DEMFI_STG_NT & 4) ? 2 : 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NIW; ++k) {
                    const int i = dw + SG_NH * k;
                    if (i >= P_NI) continue;
                    const int iy = oy0 - 1 + (lyx[k] & 255), ix = ox0 - 1 + (lyx[k] >> 8);
                    const char* g = (lyx[k] != 0xffff && iy >= 0 && iy < H && ix >= 0 && ix < W) ? base + off[k] : zeros;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, (DEMFI_STG_NT & 4) ? 2 : 0);
                }
            }
        };
        half_t* const dstp = (half_t*)sg0.dst.ptr + d->oct_ch[0];
        constexpr int NCH = 32 / SG_NH;                          // 1-KiB chunks (8 pixels x 128 B) of the 32-KiB staging per helper
        u4_t stage[NCH];
        auto stage_read = [&](int b) {
            const char* sbp = tbuf + b * P_TILE_BYTES + lane * 16;
#pragma unroll
            for (int k = 0; k < NCH; ++k) stage[k] = *(const u4_t*)(sbp + (dw + SG_NH * k) * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // in registers before this wave's DMA may overwrite the chunks
        };
        // byte offset of this lane's 16-byte piece of chunk k relative to the tile's first output pixel (loop-invariant)
        unsigned doff[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int j = dw + SG_NH * k;                        // chunk j: pixels 8j .. 8j+7 of the 8x32 tile; lane -> (pixel, physical slot)
            const int oxl = (j & 3) * 8 + (lane >> 3);
            const int q = (lane & 7) ^ ((oxl >> 1) & 7);         // logical 16-byte slot = channels 8q .. 8q+7
            doff[k] = (unsigned)(((j >> 2) * d_sy + oxl * d_sx + q * 8) * 2);
        }
        auto stage_store = [&](int bimg, int oy0, int ox0) {
            char* const obase = (char*)(dstp + bimg * d_sb + oy0 * d_sy + ox0 * d_sx);      // wave-uniform
            if (oy0 + TH <= H && ox0 + TW <= W) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    if constexpr ((DEMFI_STG_NT & 1) != 0) __builtin_nontemporal_store(stage[k], gp<u4_t>(obase + doff[k]));
                    else *gp<u4_t>(obase + doff[k]) = stage[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int j = dw + SG_NH * k;
                    if (oy0 + (j >> 2) < H && ox0 + (j & 3) * 8 + (lane >> 3) < W) *gp<u4_t>(obase + doff[k]) = stage[k];
                }
            }
        };
        const uint4* wsrc = (const uint4*)d->wpack;
        for (int i = dw; i < NTAPS * NKS * NCO; i += SG_NH)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + i * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(wlds + i * 1024), 16, 0, 0);
        issue_tile(t_first, 0);
        int buf = 0;
        int pb = 0, py = 0, px = 0;
        bool have_prev = false;
        for (int t = t_first; t < t_end; t += t_step, buf ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile t has landed (and this wave's stores of tile t-2 are out)
            TRACE_STAMP(wave, trk, 0);
            asm volatile("s_barrier" ::: "memory");             // A: tile t to the MFMA waves; their outputs of tile t-1 are staged in buffer buf^1
            TRACE_STAMP(wave, trk, 1);
            // stores first: the vmcnt(0) in front of the next barrier A then waits for the tile loads issued LAST, not for the
            // acknowledgements of stores issued late in the phase
            if (have_prev) { stage_read(buf ^ 1); stage_store(pb, py, px); }
            if (t + t_step < t_end) issue_tile(t + t_step, buf ^ 1);
#ifdef DEMFI_ABLATION
            // experiment (DEMFI_KNOB bit 6; round 4): what would STREAMING the 72 KiB of weights per tile through the helpers cost (the
            // design VERDICT r3 item 1 proposes to free LDS for a third tile buffer)?  The helpers re-issue the LDS-DMA of the resident
            // weights every tile: the same bytes land on top of themselves, results stay correct, and the helper path carries the 72
            // extra DMA instructions + 72 KiB of L2 -> LDS traffic per tile that a weight ring would add.  profiles/r04_notes.md section 6.
            if (DEMFI_KNOB_BIT(64)) {
                for (int i = dw; i < NTAPS * NKS * NCO; i += SG_NH)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + i * 64 + lane),
                                                     (__attribute__((address_space(3))) void*)(wlds + i * 1024), 16, 0, 0);
            }
#endif
            TRACE_STAMP(wave, trk, 2);
            tile_coords(t, pb, py, px);
            have_prev = true;
            asm volatile("s_barrier" ::: "memory");             // B: the MFMA waves have finished reading buffer buf and may stage into it
            ++trk;
        }
        asm volatile("s_barrier" ::: "memory");                 // F: the last tile is staged (in buffer buf^1: buf was toggled on exit)
        stage_read(buf ^ 1);
        stage_store(pb, py, px);
        return;
    }

    // ================= MFMA waves ============================================================================
    const int hi = lane >> 5;
    const int lx = lane & 31;
    const half_t* const resp = (const half_t*)sg0.res.ptr;
    const int64_t r_sx = sg0.res.sx, r_sy = sg0.res.sy, r_sb = sg0.res.sb;
    const float act_floor = sg0.act == DEMFI_ACT_RELU ? 0.0f : -__builtin_huge_valf();
    h8_t act_floor8;
#pragma unroll
    for (int j = 0; j < 8; ++j) act_floor8[j] = (half_t)act_floor;
    const int ch0 = d->oct_ch[0];
    // bias of this lane's 32 accumulator rows (MFMA-row order: element 4g + j = quad g, half hi, j), in registers for the whole
    // launch and fed to the FIRST MFMA of every accumulator as its C operand: the accumulation starts at the bias, so the
    // epilogue has no bias adds at all (the 4-wave kernel re-reads the bias from LDS per epilogue unit: 8 dependent LDS round
    // trips, ~700 cycles per tile in the phase trace).  fp32 summation order differs from "sum, then + bias" by one rounding.
    f16x_t bias16[NCO];
#pragma unroll
    for (int s = 0; s < NCO; ++s) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f4_t bq = *gcp<f4_t>(d->bias + s * 32 + g * 8 + hi * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) bias16[s][g * 4 + j] = bq[j];
        }
    }
    int boff[12];
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const int col = lx + (g >> 2);
        boff[g] = col * 128 + ((((g & 3) * 2 + hi) ^ ((col >> 1) & 7)) << 4);
    }
    const char* const wl = wlds + lane * 16;
    // staging slot of this lane's (s, m2) piece: pixel record (row, lx) of a 32-pixel-per-row image, 16-byte slot
    // q = 4s + 2m2 + hi XOR-swizzled by the column like the input tiles (conflict-free ds_write_b128 groups)
    int soff[NCO][2];
#pragma unroll
    for (int s = 0; s < NCO; ++s) {
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) soff[s][m2] = lx * 128 + (((s * 4 + m2 * 2 + hi) ^ ((lx >> 1) & 7)) << 4);
    }
    // Residual: issued in the MIDDLE of an MFMA phase (after a third of the k-loop).  At the head of the period the CU's memory pipe
    // belongs to the helper waves' stores of the previous tile and to the DMA of the next one.  Round 4: the loads issued during tile k
    // are those of tile k+1 (two register sets, the tile loop unrolled by two so that both are statically named): on the memory wall
    // the residual variant sits on (4.85 TB/s) loads issued 4 000 cycles before their use were 1 100-2 300 cycles late; a whole period
    // of lead takes that wait out of the epilogue (DEMFI_STG_RES_AHEAD 0: the tile's own residual, the round-3 schedule).
    using ResRegs = u4_t[NCO][2][2];
    constexpr bool AHEAD = RES && DEMFI_STG_RES_AHEAD != 0;
    auto tile_body = [&](const int t, const int buf, ResRegs& rreg, ResRegs& rnext) {
        int bimg, oy0, ox0;
        tile_coords(t, bimg, oy0, ox0);
        auto load_res_of = [&](ResRegs& rr, int tt) {
            if constexpr (RES) {
                int rb, ry0, rx0;
                tile_coords(tt, rb, ry0, rx0);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int oy = min(ry0 + wave * 2 + p, H - 1), oxx = min(rx0 + lx, W - 1);
                    const half_t* rp = resp + rb * r_sb + oy * r_sy + oxx * r_sx + ch0 + hi * 8;
#pragma unroll
                    for (int s = 0; s < NCO; ++s) {
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) {
                            if constexpr ((DEMFI_STG_NT & 2) != 0) rr[s][p][m2] = __builtin_nontemporal_load(gcp<u4_t>(rp + s * 32 + m2 * 16));
                            else rr[s][p][m2] = *gcp<u4_t>(rp + s * 32 + m2 * 16);
                        }
                    }
                }
            }
        };
        auto load_res = [&]() {
            // no branch inside the MFMA phase: the last tile of the walk re-reads its own residual into the idle set
            if constexpr (AHEAD) load_res_of(rnext, t + t_step < t_end ? t + t_step : t);
            else load_res_of(rreg, t);
        };
        if constexpr (DEMFI_STG_RES_AT < 0) load_res();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the staged outputs of the previous tile are in LDS
        TRACE_STAMP(wave, trk, 0);
        asm volatile("s_barrier" ::: "memory");                 // A
        TRACE_STAMP(wave, trk, 1);
        f16x_t acc[NCO][2];
        char* const tbase = tbuf + buf * P_TILE_BYTES;
        const char* tb = tbase + (wave * 2) * (P_LW * 128);
        {
            struct RowFrag { uint4 a[3][NCO]; uint4 b[4]; };
            auto load_g = [&](RowFrag& f, int g) {              // g = kx*4 + ks
                const int kx = g >> 2, ks = g & 3;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int s = 0; s < NCO; ++s) f.a[ky][s] = *(const uint4*)(wl + (((ky * 3 + kx) * NKS + ks) * NCO + s) * 1024);
                }
                const char* p0 = tb + boff[kx * 4 + ks];
#pragma unroll
                for (int r = 0; r < 4; ++r) f.b[r] = *(const uint4*)(p0 + r * (P_LW * 128));
            };
            auto mma_g = [&](const RowFrag& f, auto FIRST) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int s = 0; s < NCO; ++s) {
                        if (decltype(FIRST)::value && ky == 0) {
                            Mma<half_t>::initc(acc[s][0], f.a[ky][s], f.b[ky], bias16[s]);
                            Mma<half_t>::initc(acc[s][1], f.a[ky][s], f.b[ky + 1], bias16[s]);
                        } else {
                            Mma<half_t>::run(acc[s][0], f.a[ky][s], f.b[ky]);
                            Mma<half_t>::run(acc[s][1], f.a[ky][s], f.b[ky + 1]);
                        }
                    }
                }
            };
            auto groups = [&](bool loads) {
#pragma unroll
                for (int q = 0; q < 6 * NCO; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (loads && q < 3 * NCO + 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            RowFrag f0, f1;
            load_g(f0, 0);
            static_for<0, 6>([&](auto I) {
                constexpr int i = decltype(I)::value;
                load_g(f1, 2 * i + 1);
                mma_g(f0, std::integral_constant<bool, i == 0>{});
                groups(true);
                if constexpr (i == DEMFI_STG_RES_AT) load_res();
                if constexpr (i < 5) load_g(f0, 2 * i + 2);
                mma_g(f1, std::false_type{});
                groups(i < 5);
            });
        }
#if defined(DEMFI_TRACE) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int s = 0; s < NCO; ++s) { asm volatile("" ::"v"(acc[s][0])); asm volatile("" ::"v"(acc[s][1])); }
        TRACE_STAMP(wave, trk, 2);
#endif
        // B: every MFMA wave has consumed its reads of this tile buffer -> its first 32 KiB become the staging image of the outputs
        asm volatile("s_barrier" ::: "memory");
        TRACE_STAMP(wave, trk, 4);
        if constexpr (RES) {
#pragma unroll
            for (int s = 0; s < NCO; ++s) {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(rreg[s][q >> 1][q & 1]));
            }
        }
#pragma unroll
        for (int s = 0; s < NCO; ++s) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[s][p][(2 * m2) * 4 + j];
                        v[4 + j] = acc[s][p][(2 * m2 + 1) * 4 + j];
                    }
                    if constexpr (RES) {
                        // + residual: v_fma_mix_f32 (fp16 operand * 1.0 + fp32) = the conversion and the add in one instruction, same rounding
                        const u4_t r = rreg[s][p][m2];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[2 * q] = res_mix_lo(r[q], v[2 * q]);
                            v[2 * q + 1] = res_mix_hi(r[q], v[2 * q + 1]);
                        }
                    }
                    if constexpr (TANH) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = fast_tanh(v[j]);
                    }
                    h8_t o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
                    if constexpr (!TANH) o = __builtin_elementwise_max(o, act_floor8);
                    *(u4_t*)(tbase + (wave * 2 + p) * 4096 + soff[s][m2]) = __builtin_bit_cast(u4_t, o);
                }
            }
        }
        TRACE_STAMP(wave, trk, 3);
        ++trk;
    };
    ResRegs r_even, r_odd;
    if constexpr (AHEAD) {                                      // the first tile's residual (tile_body only fetches ahead)
        int rb, ry0, rx0;
        tile_coords(t_first, rb, ry0, rx0);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int oy = min(ry0 + wave * 2 + p, H - 1), oxx = min(rx0 + lx, W - 1);
            const half_t* rp = resp + rb * r_sb + oy * r_sy + oxx * r_sx + ch0 + hi * 8;
#pragma unroll
            for (int s = 0; s < NCO; ++s) {
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) r_even[s][p][m2] = *gcp<u4_t>(rp + s * 32 + m2 * 16);
            }
        }
    }
    for (int t = t_first; t < t_end;) {
        tile_body(t, 0, r_even, r_odd);
        t += t_step;
        if (t >= t_end) break;
        tile_body(t, 1, r_odd, r_even);
        t += t_step;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the last tile is staged
    asm volatile("s_barrier" ::: "memory");                     // F
}

static int launch_stg(const demfi_conv* h, const demfi_conv* dev, hipStream_t st)
{
    const size_t lds = 9 * 4 * 2 * 1024 + 2 * P_TILE_BYTES + 1024;
    DEMFI_LDS_ATTR((conv3x3_c64_stg_kernel<true, false>));
    DEMFI_LDS_ATTR((conv3x3_c64_stg_kernel<false, false>));
    DEMFI_LDS_ATTR((conv3x3_c64_stg_kernel<true, true>));
    DEMFI_LDS_ATTR((conv3x3_c64_stg_kernel<false, true>));
    const int total = ((h->W + TW - 1) / TW) * ((h->H + TH - 1) / TH) * h->batch;
    const int grid = total >= 256 ? 256 : total;
    const demfi_seg& sg = h->segs[h->sub_seg[0]];
    const bool res = sg.res.ptr != nullptr, th = sg.act == DEMFI_ACT_TANH;
    if (res && th)  hipLaunchKernelGGL((conv3x3_c64_stg_kernel<true, true>), dim3(grid), dim3(SG_NT), lds, st, dev);
    else if (res)   hipLaunchKernelGGL((conv3x3_c64_stg_kernel<true, false>), dim3(grid), dim3(SG_NT), lds, st, dev);
    else if (th)    hipLaunchKernelGGL((conv3x3_c64_stg_kernel<false, true>), dim3(grid), dim3(SG_NT), lds, st, dev);
    else            hipLaunchKernelGGL((conv3x3_c64_stg_kernel<false, false>), dim3(grid), dim3(SG_NT), lds, st, dev);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}


// ======================================================================================================
// Persistent 3x3 kernel for the NARROW layers (fp16, stride 1): K = 16, 32 or 64 input channels in ONE chunk made of
// up to two NHWC pieces (+ zero padding), <= 64 output channels -- Mixer conv_delta1/2, conv_blend1/2
// (DeMFInet.py:800-836) and every other layer of that shape.  These layers are HBM-bound (2-35 GFLOP on 75-180 MB), and
// the general kernel spent its time on per-tile overhead (one workgroup per tile: weight ring, VGPR-staged input,
// 9 barriers).  Same machinery as the 64-channel kernel above with the record size as a template parameter:
//   * REC = 32 / 64 / 128 bytes per pixel record (NKS = 1 / 2 / 4 k-steps); XOR swizzle of the 16-byte slot by record
//     column, chosen per REC so that the ds_read_b128 lane groups stay conflict-free;
//   * the DMA wave composes a record from the pieces: per instruction and lane a precomputed (piece, byte offset);
//   * the smaller the record the deeper the tile ring (2 / 3 / 4 buffers, 1-3 tiles in flight, counted vmcnt): a tile of
//     these layers is only ~1 us of work, much less than the HBM latency;
//   * MFMA loop pipelined per k-step (fragments two steps ahead); register epilogue as above.
// ======================================================================================================
template <int REC, int KS = 3> struct NarrowCfg {                // KS: filter size (3, or 7 for Mixer.conv_delta1)
    static constexpr int LW = TW + KS - 1, LH = TH + KS - 1, NP = LW * LH, PAD = KS / 2, NTAPS = KS * KS;
    static constexpr int NKS = REC / 32;                        // k-steps per tap
    static constexpr int SL = REC / 16;                         // 16-byte slots per record
    static constexpr int PPI = 1024 / REC;                      // records per DMA instruction
    static constexpr int NI = (NP + PPI - 1) / PPI;             // DMA instructions per tile: 43 / 22 / 11 (3x3), 17 (7x7, REC 32)
    static constexpr int TILE_BYTES = NI * 1024;
#ifndef DEMFI_N64_NBUF                                           // A/B switches of the narrow kernel's ring depth / DMA waves (same-box bench:
#define DEMFI_N64_NBUF 3                                        // 4 DMA waves for 64-byte records and 2 for 32-byte ones +0.5 %; a 4th buffer nothing)
#define DEMFI_N64_NDMA 4
#define DEMFI_N32_NDMA 2
#endif
    static constexpr int NBUF = REC == 128 ? 2 : (REC == 64 ? DEMFI_N64_NBUF : 4);
    static_assert(KS == 3 || (KS == 7 && REC == 32), "7x7: one 16-channel k-step per tap (49 KiB of resident weights)");
    // waves issuing the tile DMA (see the kernel): one wave needs NI x ~80 cycles to issue a tile
    static constexpr int NDMA = REC == 128 ? 4 : (REC == 64 ? DEMFI_N64_NDMA : (KS == 7 ? 2 : DEMFI_N32_NDMA));
    static_assert((NBUF - 1) * ((NI + NDMA - 1) / NDMA) <= 63, "a DMA wave's tiles in flight must be countable in vmcnt");
    static __device__ __forceinline__ int swz(int col) { return REC == 128 ? (col >> 1) & 7 : (REC == 64 ? (col >> 2) & 3 : (col >> 4) & 1); }
    static constexpr size_t lds_bytes(int nco) { return (size_t)NTAPS * NKS * nco * 1024 + (size_t)NBUF * TILE_BYTES + 1024; }
};

struct NarrowFrag { uint4 a[2], b0, b1; };

// EPI: 0 = one NHWC fp16 destination, 1 = the same + residual, 2 = THIN: planar fp32 destinations / residuals routed per
// octet (Dec_last2, Dec_last2_2, flow_occ.conv2, w_gen_2: <= 32 packed couts, NCO == 1)
// NDMA: waves that issue the LDS-DMA of a tile (instruction i belongs to DMA wave i % NDMA).  One wave needs 43 x ~80 cycles
// just to ISSUE a 128-byte-record tile; the thin-output layers (little MFMA work per tile, two tile buffers) are bound by
// exactly that latency, so they use two.
// NOCT (THIN only): number of live 8-cout octets (they are the first NOCT ones).  The thin-output layers have 1-3 (Dec_last2 1,
// flow_occ.conv2 / dec3's planes 2, Dec_last2_2 3); round 2 walked all four unconditionally: 32 scalar residual loads and four
// dependent LDS bias reads per tile whatever the layer -- the phase trace (profiles/r03_notes.md) shows 1 860 + 2 810 of a 8 260-cycle
// period of Dec_last2 there.
// PACK (THIN only): the layer also writes the packed fp16 copy of its planes (demfi_conv.pack) -- its own instantiation: the extra
// pointers cost the plain thin layers 5-9 % when they were a run-time option (Dec_last2_2 0.485 -> 0.52 ms per 7 t, same box)
// REGW (THIN, 3x3): the layer's weight fragments (9 taps x NKS k-steps, one 32-cout subtile: 18 / 36 x 4 registers) live in the MFMA
// waves' REGISTERS for the whole launch instead of being re-read from LDS by every wave for every tile.  The thin layers' MFMA phase is
// LDS-read bound (336 KiB of fragment reads per 8 x 32 tile of Dec_last2 for 36 MFMAs per wave: profiles/r03_notes.md section 4); the A
// fragments are 43 % of those reads.  Round 4.
// 4 x 4 transpose inside a quad of lanes (two rounds of DPP exchanges): in: a[j] = element j of this lane's row; out: a[k] = element
// (this lane's index in its quad) of the row of quad lane k.  The thin epilogue uses it to turn "4 channels of one pixel" (the MFMA
// accumulator layout) into "4 consecutive pixels of one channel" = one 16-byte access to a planar fp32 tensor.
__device__ __forceinline__ void quad_transpose4(float (&a)[4], int lane)
{
    auto dpp = [](float v, auto CTRL) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(CTRL)::value, 0xF, 0xF, false));
    };
    const bool b0 = lane & 1, b1 = lane & 2;
    float p[4], y[4], q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = dpp(a[j], std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]: lane ^ 1
    y[0] = b0 ? p[1] : a[0]; y[1] = b0 ? a[1] : p[0]; y[2] = b0 ? p[3] : a[2]; y[3] = b0 ? a[3] : p[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = dpp(y[j], std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]: lane ^ 2
    a[0] = b1 ? q[2] : y[0]; a[1] = b1 ? q[3] : y[1]; a[2] = b1 ? y[2] : q[0]; a[3] = b1 ? y[3] : q[1];
}
#ifndef DEMFI_THIN_VEC
#define DEMFI_THIN_VEC 1                                         // 0: A/B builds without the quad-transposed 16-byte epilogue accesses
#endif
#ifndef DEMFI_THIN_REGW
#define DEMFI_THIN_REGW 1
#endif
#ifndef DEMFI_THIN_REGW_G128
#define DEMFI_THIN_REGW_G128 6
#endif
template <int NCO, int REC, int EPI, int KS = 3, int NDMA = NarrowCfg<REC, KS>::NDMA, int NOCT = 4, bool PACK = false>
__global__ __launch_bounds__(NT + 64 * NDMA, 1) void conv3x3_narrow_persist_kernel(const demfi_conv* __restrict__ d)
{
    constexpr bool RES = EPI == 1, THIN = EPI == 2;
    constexpr bool REGW = THIN && KS == 3 && NCO == 1 && DEMFI_THIN_REGW != 0;
    // (kx, k-step) groups whose three ky fragments are register resident: all 6 of a 64-byte-record layer (18 fragments, 72 registers),
    // 6 of the 12 of a 128-byte-record layer (all 36 = 144 registers spill in the 256-register budget of this 8-wave workgroup); the
    // other groups keep reading the LDS copy
    // Measured (profiles/r04_notes.md section 8, same box, alternating libraries): Dec_last2 (128-byte records, one live octet, 6 of 12 groups
    // resident) 0.943 -> 0.901 ms; Dec_last2_2 (three octets, 4 groups) 0.471 -> 0.495 and flow_occ.conv2 (64-byte records, all 6 groups)
    // 0.233 -> 0.250: SLOWER -- the thin layers are not bound by the A-fragment LDS reads, and the extra registers cost more than
    // the reads save.  Enabled only where it paid.
    constexpr int RG = (REGW && REC == 128 && NOCT == 1 && !PACK) ? DEMFI_THIN_REGW_G128 : 0;
    constexpr bool WLDS = RG < (REC / 32) * 3 || !REGW;          // the LDS copy of the weights is (still) needed
    static_assert(!PACK || THIN, "packed copy: thin epilogue only");
    static_assert(!THIN || NCO == 1, "thin epilogue: one 32-cout subtile");
    using Cfg = NarrowCfg<REC, KS>;
    constexpr int P_LW = Cfg::LW, P_NP = Cfg::NP, PAD = Cfg::PAD;      // shadow the 3x3 constants of the 64-channel kernel
    constexpr int NKS = Cfg::NKS, SL = Cfg::SL, NI = Cfg::NI, NBUF = Cfg::NBUF, TILE_BYTES = Cfg::TILE_BYTES;
    constexpr int NSTEP = Cfg::NTAPS * NKS;
    constexpr int WBYTES = NSTEP * NCO * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d->H, W = d->W;
    const int tiles_x = (W + TW - 1) / TW;
    const int tiles_y = (H + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int total = tiles_img * d->batch;
    char* const wlds = smem;
    char* const tbuf = smem + WBYTES;
    const int G = gridDim.x;
    int t_first, t_end, t_step;
    if ((G & 7) == 0 && total >= G) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        const int lo = xcd * q + min(xcd, r);
        t_first = lo + idx;
        t_end = lo + q + (xcd < r ? 1 : 0);
        t_step = G >> 3;
    } else {
        t_first = blockIdx.x;
        t_end = total;
        t_step = G;
    }
    if (t_first >= t_end) return;                               // uniform per workgroup
    const int n_tiles = (t_end - t_first + t_step - 1) / t_step;
    auto tile_coords = [&](int t, int& bimg, int& oy0, int& ox0) {
        bimg = t / tiles_img;
        const int rem = t - bimg * tiles_img;
        const int ty = rem / tiles_x;
        oy0 = ty * TH;
        ox0 = (rem - ty * tiles_x) * TW;
    };

    if (wave >= 4) {
        // ================= DMA wave(s) =======================================================================
        if (DEMFI_KNOB_BIT(1)) __builtin_amdgcn_s_setprio(3);
        const int dw = wave - 4;                                 // this wave issues instructions i with i % NDMA == dw
        constexpr int NIW = NI / NDMA;                           // instructions per tile and wave, rounded DOWN (vmcnt waits err on the safe side)
        // the (at most two) real pieces of the chunk; everything else of the record is zero padding
        const demfi_chunk& ch = d->chunks[0];
        const char* src[2] = {nullptr, nullptr};
        int64_t psx[2] = {0, 0}, psy[2] = {0, 0}, psb[2] = {0, 0};
        int pb0[2] = {0, 0}, pb1[2] = {0, 0};                    // byte range of the piece inside the record
        int nreal = 0;
        for (int k = 0; k < ch.n_pieces; ++k) {
            const demfi_piece& pc = d->pieces[ch.first_piece + k];
            if (pc.v.ptr == nullptr || nreal == 2) continue;
            src[nreal] = (const char*)pc.v.ptr;
            psx[nreal] = pc.v.sx * 2; psy[nreal] = pc.v.sy * 2; psb[nreal] = pc.v.sb * 2;
            pb0[nreal] = pc.lds_ch * 2; pb1[nreal] = (pc.lds_ch + pc.nch) * 2;
            ++nreal;
        }
        const char* const zeros = (const char*)d->zero_page;
        // instruction i covers records PPI*i ..; lane -> (record PPI*i + lane/SL, physical slot lane%SL)
        int off[NI], meta[NI];                                    // meta = row | column << 8 | piece << 16 (piece 2 = zeros)
        bool any_other = false;                                   // some lane of some instruction is NOT a plain piece-0 slot
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int px = i * Cfg::PPI + lane / SL;
            const int pxc = min(px, P_NP - 1);                   // records past the tile (last instruction): re-read the last record, never consumed
            const int ly = pxc / P_LW;
            const int lxx = pxc - ly * P_LW;
            const int byte = (((lane & (SL - 1)) ^ Cfg::swz(lxx)) << 4);      // logical slot held by this physical slot
            int sel = 2;
            if (byte >= pb0[0] && byte < pb1[0]) sel = 0;
            else if (byte >= pb0[1] && byte < pb1[1]) sel = 1;
            const int pi = sel == 1 ? 1 : 0;
            off[i] = (int)(ly * psy[pi] + lxx * psx[pi]) + byte - pb0[pi];
            meta[i] = px < P_NP ? (ly | (lxx << 8) | (sel << 16)) : (0xffff | (2 << 16));
            any_other = any_other || sel != 0;
        }
        // SIMPLE layers (one real piece that fills the whole record: Dec_last2*, flow_occ.conv2, dec3's planes, ...): an interior tile
        // is "uniform base + precomputed lane offset" per instruction -- 2 VALU instead of ~10 (select between two pieces / the zero
        // page, bounds).  The DMA waves are younger than the MFMA waves and get few issue slots (phase trace: 4 500 cycles for the
        // 11 instructions of a wave), and with a short MFMA phase their issue time IS the tile period.
        const bool simple = __builtin_amdgcn_readfirstlane(__ballot(any_other) == 0 ? 1 : 0) != 0;
        auto issue_tile = [&](int k) {
            int bimg, oy0, ox0;
            tile_coords(t_first + k * t_step, bimg, oy0, ox0);
            const char* base0 = src[0] + (int64_t)bimg * psb[0] + (int64_t)(oy0 - PAD) * psy[0] + (int64_t)(ox0 - PAD) * psx[0];
            const char* base1 = src[1] + (int64_t)bimg * psb[1] + (int64_t)(oy0 - PAD) * psy[1] + (int64_t)(ox0 - PAD) * psx[1];
            char* dst = tbuf + (k % NBUF) * TILE_BYTES;
            const bool interior = oy0 >= PAD && oy0 + TH + PAD <= H && ox0 >= PAD && ox0 + TW + PAD <= W;
            if (simple && interior) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if (NDMA > 1 && (i % NDMA) != dw) continue;  // wave-uniform
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base0 + off[i]),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (NDMA > 1 && (i % NDMA) != dw) continue;      // wave-uniform
                // the lane's (line, column, piece) word is made opaque per tile: otherwise the compiler hoists the lane MASKS of the
                // comparisons below out of the tile loop -- ~15 SGPR pairs per DMA instruction, 170-370 of them spilled to VGPR lanes and
                // read back with v_readlane + wait states on every tile (round 4: .sgpr_spill_count of the thin instantiations)
                int mt = meta[i];
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(mt));
#endif
                const int sel = mt >> 16;
                const int iy = oy0 - PAD + (mt & 255), ix = ox0 - PAD + ((mt >> 8) & 255);
                const bool ok = sel != 2 && (mt & 0xffff) != 0xffff && (interior || (iy >= 0 && iy < H && ix >= 0 && ix < W));
                const char* g = ok ? (sel == 1 ? base1 : base0) + off[i] : zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
        };
        const uint4* wsrc = (const uint4*)d->wpack;
        if constexpr (WLDS) {
            for (int i = dw; i < NSTEP * NCO; i += NDMA)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + i * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(wlds + i * 1024), 16, 0, 0);
        }
        for (int k = 0; k < NBUF - 1 && k < n_tiles; ++k) issue_tile(k);
        for (int k = 0; k < n_tiles; ++k) {
            // tiles k+1 .. k+NBUF-2 (those that exist) may stay in flight; loads retire in order
            const int ahead = min(NBUF - 2, n_tiles - 1 - k);
            if (ahead >= 2)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIW <= 63 ? 2 * NIW : 0) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW <= 63 ? NIW : 0) : "memory");
            else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TRACE_STAMP(wave, k, 0);
            __syncthreads();                                    // hand tile k to the MFMA waves
            TRACE_STAMP(wave, k, 1);
            // ring slot of tile k+NBUF-1 = slot of tile k-1: every MFMA wave finished reading it before this barrier
            if (k + NBUF - 1 < n_tiles) issue_tile(k + NBUF - 1);
            TRACE_STAMP(wave, k, 2);
        }
        return;
    }

    // ================= MFMA waves ============================================================================
    const int hi = lane >> 5;
    const int lx = lane & 31;
    const demfi_seg& sg0 = d->segs[THIN ? 0 : d->sub_seg[0]];
    half_t* const dstp = (half_t*)sg0.dst.ptr;
    const half_t* const resp = (const half_t*)sg0.res.ptr;
    const int64_t d_sx = sg0.dst.sx, d_sy = sg0.dst.sy, d_sb = sg0.dst.sb;
    const int64_t r_sx = sg0.res.sx, r_sy = sg0.res.sy, r_sb = sg0.res.sb;
    const float act_floor = sg0.act == DEMFI_ACT_RELU ? 0.0f : -__builtin_huge_valf();
    const int ch0 = d->oct_ch[0];
    float* const bias_lds = (float*)(tbuf + NBUF * TILE_BYTES);
    if (tid < NCO * 32) bias_lds[tid] = d->bias[tid];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int boff[KS * NKS];                                         // [kx*NKS + ks]: record (lx + kx) + swizzled 16-byte slot
#pragma unroll
    for (int g = 0; g < KS * NKS; ++g) {
        const int col = lx + g / NKS;
        boff[g] = col * REC + ((((g % NKS) * 2 + hi) ^ Cfg::swz(col)) << 4);
    }
    const char* const wl = wlds + lane * 16;
    uint4 areg[RG > 0 ? RG * 3 : 1];                              // REGW: fragment (ky, group g < RG) = areg[g * 3 + ky], indexed by constants only
    if constexpr (REGW) {
        const char* wg = (const char*)d->wpack + lane * 16;
        static_for<0, RG * 3>([&](auto I_) {
            constexpr int i = decltype(I_)::value, g = i / 3, ky = i % 3, kx = g / NKS, ks = g % NKS;
            areg[i] = ld_global16(wg + ((ky * 3 + kx) * NKS + ks) * 1024);
        });
    }
    // ---- THIN: per octet g (= accumulator quad g) the planar destination / residual of this lane's 4 channels -------
    // packed cout of accumulator element (g, j) of this lane: 8g + 4hi + j; valid when 4hi + j < oct_n[g]
    float* t_dst[4];
    const float* t_res[4];
    int t_on[4], t_act[4], t_nq[4], t_rmul[4];
    int64_t t_dsb[4], t_rsb[4];
    int64_t t_dsc[4], t_dsx[4], t_dsy[4], t_rsc[4], t_rsx[4], t_rsy[4];   // element strides of the planar views (any: e.g. the parity views of dec3)
    f4_t t_bias[4];                                              // bias of this lane's quad of octet g, in registers (was: LDS read per tile)
    if constexpr (THIN) {
#pragma unroll
        for (int g = 0; g < NOCT; ++g) {
            t_bias[g] = *gcp<f4_t>(d->bias + g * 8 + 4 * hi);
            t_on[g] = d->oct_n[g];
            const demfi_seg& sg = d->segs[d->oct_seg[g]];
            t_act[g] = sg.act;
            t_nq[g] = min(max(t_on[g] - 4 * hi, 0), 4);
            const int c0 = d->oct_ch[g] + (t_nq[g] > 0 ? 4 * hi : 0);       // lanes without a valid channel shadow channel 0 (never stored)
            t_dsc[g] = sg.dst.sc; t_dsx[g] = sg.dst.sx; t_dsy[g] = sg.dst.sy;
            t_rsc[g] = sg.res.sc; t_rsx[g] = sg.res.sx; t_rsy[g] = sg.res.sy;
            t_dst[g] = (float*)sg.dst.ptr + c0 * t_dsc[g];
            // no residual (or an empty octet): the prefetch below reads the zero page with all strides multiplied by 0, so
            // that it stays unconditional (conditional loads leave register copies + an s_waitcnt in front of the MFMAs)
            const bool hasres = t_on[g] > 0 && sg.res.ptr != nullptr;
            t_res[g] = hasres ? (const float*)sg.res.ptr + c0 * t_rsc[g] : (const float*)d->zero_page;
            t_rmul[g] = hasres ? 1 : 0;
            t_dsb[g] = sg.dst.sb;
            t_rsb[g] = sg.res.sb;
        }
    }
    // ---- THIN, vector accesses (round 4): after a 4 x 4 transpose inside each lane quad, lane (quad lane q4) holds channel 4 hi + q4 of
    // its octet for the quad's 4 consecutive pixels: ONE 16-byte residual load and ONE 16-byte store per (octet, row) and lane instead
    // of four 4-byte ones -- the thin epilogue is bound by the NUMBER of VMEM instructions its waves issue (phase trace: 4 300 cycles
    // for the 36 accesses per tile and wave of Dec_last2_2).  Needs unit-stride, 16-byte aligned planes and a tile inside the image;
    // tiles / layers that do not qualify (ragged edges, the parity views of dec3, an active uint8 sink) take the scalar accesses.
    const int q4 = lx & 3;
    float* t_dstq[4];
    const float* t_resq[4];
    bool tv_ok = THIN && DEMFI_THIN_VEC != 0;
    if constexpr (THIN) {
        auto al16 = [](const void* pp, int64_t a, int64_t b, int64_t c) { return (((uintptr_t)pp) & 15) == 0 && ((a | b | c) & 3) == 0; };
#pragma unroll
        for (int g = 0; g < NOCT; ++g) {
            const demfi_seg& sg = d->segs[d->oct_seg[g]];
            const int qq = min(q4, max(t_nq[g] - 1, 0));         // lanes past the last valid channel shadow it (loaded, never stored)
            t_dstq[g] = t_dst[g] + qq * t_dsc[g];
            t_resq[g] = t_res[g] + qq * t_rsc[g] * t_rmul[g];
            if (t_on[g] > 0)
                tv_ok = tv_ok && t_dsx[g] == 1 && al16(sg.dst.ptr, t_dsc[g], t_dsy[g], t_dsb[g]) &&
                        (t_rmul[g] == 0 || (t_rsx[g] == 1 && al16(sg.res.ptr, t_rsc[g], t_rsy[g], t_rsb[g])));
        }
    }
    // optional uint8 sink (demfi_u8_sink, read at run time so that one captured graph serves every destination): octet g
    // = one 3-channel frame segment whose channels all sit in the hi == 0 lane's quad
    // Batch image b uses the record DEMFI_U8_SINK_STRIDE * b bytes behind it (the batched per-t plan: one record per context);
    // a workgroup's tile band crosses an image boundary once or twice per launch, so the record is re-read only then.
    unsigned char* s_dst[4] = {nullptr, nullptr, nullptr, nullptr};
    int s_h = 0, s_w = 0, s_img = -1;
    // optional packed copy (demfi_conv.pack): this lane's group of octet g goes to channels pack_oct_ch[g] + 4 hi .. of the NHWC record
    half_t* pk_dst[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t pk_sx = 0, pk_sy = 0, pk_sb = 0;
    if constexpr (PACK) {
        if (d->pack.ptr != nullptr) {
            pk_sx = d->pack.sx; pk_sy = d->pack.sy; pk_sb = d->pack.sb;
#pragma unroll
            for (int g = 0; g < NOCT; ++g)
                if (d->pack_oct_ch[g] >= 0 && t_nq[g] > 0) pk_dst[g] = (half_t*)d->pack.ptr + d->pack_oct_ch[g] + 4 * hi;
        }
    }
    int slot = 0;
    for (int k = 0; k < n_tiles; ++k) {
        int bimg, oy0, ox0;
        tile_coords(t_first + k * t_step, bimg, oy0, ox0);
        if constexpr (THIN) {
            if (d->u8_sink != nullptr && bimg != s_img) {        // wave-uniform
                s_img = bimg;
                const demfi_u8_sink* sk = (const demfi_u8_sink*)((const char*)d->u8_sink + (int64_t)bimg * DEMFI_U8_SINK_STRIDE);
                const bool on = sk->iter == d->u8_iter;
                s_h = sk->h; s_w = sk->w;
#pragma unroll
                for (int g = 0; g < NOCT; ++g) s_dst[g] = (on && t_on[g] == 3 && d->oct_ch[g] == 0) ? sk->frame[d->oct_seg[g]] : nullptr;
            }
        }
        u4_t rreg[NCO][2][2];
        float tr[4][2][4];                                      // THIN: residual [octet][row][j], prefetched like rreg
        // vector accesses for this tile?  (wave-uniform; an active sink keeps the per-pixel layout for its byte stores)
        bool tvec = tv_ok && ox0 + TW <= W;
        if constexpr (THIN) {
#pragma unroll
            for (int g = 0; g < NOCT; ++g) tvec = tvec && s_dst[g] == nullptr;
        }
        if constexpr (THIN) {
            if (tvec) {
#pragma unroll
                for (int g = 0; g < NOCT; ++g) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {                // [octet][row][pixel of the quad]: channel 4 hi + q4, 16 bytes
                        const int oy = min(oy0 + wave * 2 + p, H - 1);
                        const float* rp = t_resq[g] + (bimg * t_rsb[g] + (int64_t)oy * t_rsy[g] + ox0 + (lx & ~3)) * t_rmul[g];
                        const f4_t rv = *gcp<f4_t>(rp);
#pragma unroll
                        for (int j = 0; j < 4; ++j) tr[g][p][j] = rv[j];
                    }
                }
            } else {
#pragma unroll
            for (int g = 0; g < NOCT; ++g) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int oy = min(oy0 + wave * 2 + p, H - 1), oxx = min(ox0 + lx, W - 1);
                    const float* rp = t_res[g] + (bimg * t_rsb[g] + (int64_t)oy * t_rsy[g] + (int64_t)oxx * t_rsx[g]) * t_rmul[g];
#pragma unroll
                    for (int j = 0; j < 4; ++j)                 // invalid j of this lane: re-read its first channel (value unused)
                        tr[g][p][j] = *gcp<float>(rp + (j < t_nq[g] ? j : 0) * t_rsc[g] * t_rmul[g]);
                }
            }
            }
        }
        if constexpr (RES) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int oy = min(oy0 + wave * 2 + p, H - 1), oxx = min(ox0 + lx, W - 1);
                const half_t* rp = resp + bimg * r_sb + oy * r_sy + oxx * r_sx + ch0 + hi * 8;
#pragma unroll
                for (int s = 0; s < NCO; ++s) {
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) rreg[s][p][m2] = *gcp<u4_t>(rp + s * 32 + m2 * 16);
                }
            }
        }
        TRACE_STAMP(wave, k, 0);
        asm volatile("s_barrier" ::: "memory");                 // tile k is in ring slot `slot`
        TRACE_STAMP(wave, k, 1);
        f16x_t acc[NCO][2];
#pragma unroll
        for (int s = 0; s < NCO; ++s) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[s][0][i] = 0.0f; acc[s][1][i] = 0.0f; }
        }
        const char* tb = tbuf + slot * TILE_BYTES + (wave * 2) * (P_LW * REC);
        slot = slot == NBUF - 1 ? 0 : slot + 1;
        if constexpr (KS == 3) {
            // (kx, k-step) groups with the three ky taps inside, like the 64 -> 64 kernel: output rows p = 0, 1 and ky = 0..2 touch the
            // four input rows p + ky at one column offset, so 4 row fragments + 3*NCO weight fragments feed 6*NCO MFMAs (round 2: one
            // (tap, k-step) at a time = 2 + NCO reads per 2*NCO MFMAs with a scheduling fence per step; the phase trace shows 3 400
            // cycles for the 36 MFMAs of the 64 -> 3 layers).  The next group's reads are interleaved 1:1 with this group's MFMAs.
            constexpr int NG = 3 * NKS;                         // groups: g = kx*NKS + ks
            struct RowFragN { uint4 a[3][NCO]; uint4 b[4]; };                   // groups g < RG: the A fragments are read straight from areg (a unused)
            auto load_g = [&](RowFragN& f, auto G_) {
                constexpr int g = decltype(G_)::value;
                constexpr int kx = g / NKS, ks = g % NKS;
                if constexpr (g >= RG) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                        for (int s = 0; s < NCO; ++s) f.a[ky][s] = *(const uint4*)(wl + ((((ky * 3 + kx) * NKS + ks) * NCO) + s) * 1024);
                    }
                }
                const char* p0 = tb + boff[kx * NKS + ks];
#pragma unroll
                for (int r = 0; r < 4; ++r) f.b[r] = *(const uint4*)(p0 + r * (P_LW * REC));
            };
            auto mma_g = [&](const RowFragN& f, auto G_) {
                constexpr int g = decltype(G_)::value;
                constexpr int kx = g / NKS, ks = g % NKS;
                static_for<0, 3>([&](auto KY_) {
                    constexpr int ky = decltype(KY_)::value;
#pragma unroll
                    for (int s = 0; s < NCO; ++s) {
                        uint4 a;
                        if constexpr (g < RG) a = areg[g * 3 + ky]; else a = f.a[ky][s];
                        Mma<half_t>::run(acc[s][0], a, f.b[ky]);
                        Mma<half_t>::run(acc[s][1], a, f.b[ky + 1]);
                    }
                });
            };
            auto groups = [&](auto NR_) {                       // NR_: ds_reads of the group being prefetched to interleave with this group's MFMAs (0: none)
                constexpr int nr = decltype(NR_)::value;
#pragma unroll
                for (int q = 0; q < 6 * NCO; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (nr > 6 * NCO) { if (q == 0) __builtin_amdgcn_sched_group_barrier(0x100, nr - 6 * NCO + 1, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    else if (q < nr) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // fragments TWO groups ahead: a group is only 6*NCO MFMAs (192 cycles for NCO = 1), less than an LDS round trip under load
            RowFragN f[3];
            load_g(f[0], std::integral_constant<int, 0>{});
            if constexpr (NG > 1) load_g(f[1], std::integral_constant<int, 1>{});
            static_for<0, NG>([&](auto G_) {
                constexpr int g = decltype(G_)::value;
                if constexpr (g + 2 < NG) load_g(f[(g + 2) % 3], std::integral_constant<int, g + 2>{});
                mma_g(f[g % 3], G_);
                groups(std::integral_constant<int, (g + 2 < NG) ? (g + 2 < RG ? 4 : 3 * NCO + 4) : 0>{});
            });
        } else {
            auto load_step = [&](NarrowFrag& f, int g) {        // g = tap*NKS + ks
                const int tap = g / NKS, ks = g % NKS;
                const int ky = tap / KS, kx = tap % KS;
#pragma unroll
                for (int s = 0; s < NCO; ++s) f.a[s] = *(const uint4*)(wl + (g * NCO + s) * 1024);
                const char* p0 = tb + boff[kx * NKS + ks];
                f.b0 = *(const uint4*)(p0 + ky * (P_LW * REC));
                f.b1 = *(const uint4*)(p0 + (ky + 1) * (P_LW * REC));
            };
            NarrowFrag f[3];                                    // fragments two k-steps ahead of the MFMAs
            load_step(f[0], 0);
            if constexpr (NSTEP > 1) load_step(f[1], 1);
            static_for<0, NSTEP>([&](auto ST) {
                constexpr int st = decltype(ST)::value;
                if constexpr (st + 2 < NSTEP) load_step(f[(st + 2) % 3], st + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < NCO; ++s) {
                    Mma<half_t>::run(acc[s][0], f[st % 3].a[s], f[st % 3].b0);
                    Mma<half_t>::run(acc[s][1], f[st % 3].a[s], f[st % 3].b1);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
#if defined(DEMFI_TRACE) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int s = 0; s < NCO; ++s) { asm volatile("" ::"v"(acc[s][0])); asm volatile("" ::"v"(acc[s][1])); }
        TRACE_STAMP(wave, k, 2);
#endif
        if constexpr (THIN) {
#pragma unroll
            for (int g = 0; g < NOCT; ++g) {                    // retire the prefetch here (see the 64-channel kernel)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(tr[g][q >> 2][q & 3]));
            }
            TRACE_STAMP(wave, k, 4);                            // residual prefetch retired (vmcnt wait over)
            if (tvec) {                                         // wave-uniform
                // Vector accesses, all (octet, row) units as ONE straight-line block (independent chains: the lone MFMA wave of a SIMD has
                // nothing else to hide VALU / DPP latency with): (acc + bias) of this lane's 4 channels -> quad transpose -> 4 pixels of
                // channel 4 hi + q4, + the residual of those 4 pixels (same two roundings per value as the scalar order), activation, ONE
                // 16-byte store.  The packed copy wants the per-pixel layout back: a second transpose.
                float vv[NOCT][2][4];
#pragma unroll
                for (int g = 0; g < NOCT; ++g) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) vv[g][p][j] = acc[0][p][g * 4 + j] + t_bias[g][j];
                    }
                }
#pragma unroll
                for (int g = 0; g < NOCT; ++g) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) quad_transpose4(vv[g][p], lane);
                }
#pragma unroll
                for (int g = 0; g < NOCT; ++g) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) vv[g][p][j] = vv[g][p][j] + tr[g][p][j];
                    }
                    apply_act_n<4>(vv[g][0], t_act[g]);
                    apply_act_n<4>(vv[g][1], t_act[g]);
                }
#pragma unroll
                for (int g = 0; g < NOCT; ++g) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int oyv = oy0 + wave * 2 + p;
                        if (oyv < H && q4 < t_nq[g]) {
                            f4_t o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = vv[g][p][j];
                            *gp<f4_t>(t_dstq[g] + bimg * t_dsb[g] + (int64_t)oyv * t_dsy[g] + ox0 + (lx & ~3)) = o;
                        }
                    }
                }
                if constexpr (PACK) {
#pragma unroll
                    for (int g = 0; g < NOCT; ++g) {
                        if (pk_dst[g] == nullptr) continue;     // depends on hi only: uniform inside a lane quad (the DPP exchange stays inside quads)
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            quad_transpose4(vv[g][p], lane);    // lanes without a valid channel carry don't-care values: only j < t_nq is used
                            const int oyv = oy0 + wave * 2 + p;
                            if (oyv < H) {
                                h4_t o;
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[j] = j < t_nq[g] ? (half_t)vv[g][p][j] : (half_t)0.0f;
                                *gp<h4_t>(pk_dst[g] + bimg * pk_sb + (int64_t)oyv * pk_sy + (int64_t)(ox0 + lx) * pk_sx) = o;
                            }
                        }
                    }
                }
                TRACE_STAMP(wave, k, 3);
                continue;
            }
#pragma unroll
            for (int g = 0; g < NOCT; ++g) {
                if (t_on[g] == 0) continue;                     // wave-uniform
                const f4_t bq = t_bias[g];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (acc[0][p][g * 4 + j] + bq[j]) + tr[g][p][j];   // + 0 without a residual
                    apply_act_n<4>(v, t_act[g]);
                    const int oy = oy0 + wave * 2 + p, oxx = ox0 + lx;
                    if (s_dst[g] != nullptr) {                  // wave-uniform: crop + denorm255 + uint8 truncation instead of the fp32 store
                        if (hi == 0 && oy < s_h && oxx < s_w) {
                            unsigned char* bp = s_dst[g] + ((int64_t)oy * s_w + oxx) * 3;
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                double q = ((double)v[j] + 1.0) / 2.0;          // denorm255_np on the float64 copy (utils.py:718-721)
                                q = q < 0.0 ? 0.0 : (q > 1.0 ? 1.0 : q);
                                *gp<unsigned char>(bp + j) = (unsigned char)(q * 255.0);   // .astype(np.uint8), main.py:1165-1178
                            }
                        }
                        continue;
                    }
                    if (oy < H && oxx < W) {
                        float* dp = t_dst[g] + bimg * t_dsb[g] + (int64_t)oy * t_dsy[g] + (int64_t)oxx * t_dsx[g];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < t_nq[g]) *gp<float>(dp + j * t_dsc[g]) = v[j];
                        if (PACK && pk_dst[g] != nullptr) {     // lane-divergent only through hi (lanes without a valid channel do not write)
                            h4_t o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = j < t_nq[g] ? (half_t)v[j] : (half_t)0.0f;
                            *gp<h4_t>(pk_dst[g] + bimg * pk_sb + (int64_t)oy * pk_sy + (int64_t)oxx * pk_sx) = o;
                        }
                    }
                }
            }
            TRACE_STAMP(wave, k, 3);
            continue;
        }
        if constexpr (RES) {
#pragma unroll
            for (int s = 0; s < NCO; ++s) {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(rreg[s][q >> 1][q & 1]));
            }
        }
#pragma unroll
        for (int s = 0; s < NCO; ++s) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const f4_t b0 = *(const f4_t*)(bias_lds + s * 32 + (2 * m2) * 8 + hi * 4);       // bias in MFMA-row order
                const f4_t b1 = *(const f4_t*)(bias_lds + s * 32 + (2 * m2 + 1) * 8 + hi * 4);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {              // cout_perm: quads 2*m2, 2*m2+1 of this lane = channels 16*m2 + 8*hi + 0..7
                        v[j] = acc[s][p][(2 * m2) * 4 + j] + b0[j];
                        v[4 + j] = acc[s][p][(2 * m2 + 1) * 4 + j] + b1[j];
                    }
                    if constexpr (RES) {
                        const h8_t r = __builtin_bit_cast(h8_t, rreg[s][p][m2]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += (float)r[j];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], act_floor);
                    const int oy = oy0 + wave * 2 + p, oxx = ox0 + lx;
                    if (oy < H && oxx < W)
                        store8<half_t>(dstp + bimg * d_sb + oy * d_sy + oxx * d_sx + ch0 + s * 32 + m2 * 16 + hi * 8, v);
                }
            }
        }
    }
}

template <int NCO, int REC, int KS = 3>
int launch_narrow(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, bool thin)
{
    const size_t lds = NarrowCfg<REC, KS>::lds_bytes(NCO);
    DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<NCO, REC, 1, KS>));
    DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<NCO, REC, 0, KS>));
    if constexpr (NCO == 1) {
        DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<1, REC, 2, KS>));
    }
    const int total = ((h->W + TW - 1) / TW) * ((h->H + TH - 1) / TH) * h->batch;
    const int grid = total >= 256 ? 256 : total;
    if (thin) {
        if constexpr (NCO == 1) {
            constexpr int ND = NarrowCfg<REC, KS>::NDMA;
            int noct = 0;                                        // live octets must be the leading ones for the specialised instantiations
            while (noct < 4 && h->oct_n[noct] > 0) ++noct;
            for (int g = noct; g < 4; ++g) if (h->oct_n[g] > 0) noct = 4;
            DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 1>));
            DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 2>));
            DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 3>));
            const dim3 blk(NT + 64 * ND);
            if (h->pack.ptr != nullptr) {                        // packed copy: the deltas' producers have 1 or 2 live octets
                DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 1, true>));
                DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 2, true>));
                DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 4, true>));
                if (noct == 1)      hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 1, true>), dim3(grid), blk, lds, st, dev);
                else if (noct == 2) hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 2, true>), dim3(grid), blk, lds, st, dev);
                // round 6: two column parities of dec3's flow / occlusion planes in one launch (4 live octets, each with its own piece of the record)
                else if (noct == 4) hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 4, true>), dim3(grid), blk, lds, st, dev);
                else return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: packed copy needs 1, 2 or 4 live octets, got %d", noct);
            } else
            if (noct == 1)      hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 1>), dim3(grid), blk, lds, st, dev);
            else if (noct == 2) hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 2>), dim3(grid), blk, lds, st, dev);
            else if (noct == 3) hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<1, REC, 2, KS, ND, 3>), dim3(grid), blk, lds, st, dev);
            else                hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<1, REC, 2, KS>), dim3(grid), blk, lds, st, dev);
        } else
            return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: thin epilogue needs nco == 1");
    } else if (h->segs[h->sub_seg[0]].res.ptr != nullptr)
        hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<NCO, REC, 1, KS>), dim3(grid), dim3(NT + 64 * NarrowCfg<REC, KS>::NDMA), lds, st, dev);
    else
        hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<NCO, REC, 0, KS>), dim3(grid), dim3(NT + 64 * NarrowCfg<REC, KS>::NDMA), lds, st, dev);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}


// ======================================================================================================
// Persistent kernel for the SepConvGRU convolutions (DeMFInet.py:838-857): fp16, 1x5 or 5x1 filter, input = two
// NHWC pieces of 64 channels (h | x resp. r*h | x), 64 output channels per workgroup (convq: 64 couts; the fused
// convz|convr launch: 128 couts = two workgroup "halves", the parity of the work item selects z or r).
// Built from the parts of the 3x3 kernel above (resident weights in LDS, DMA wave, XOR-swizzled records, register
// epilogue, raw barriers), arranged for K = 128 and a 5-tap 1-D filter:
//   * the tile is 32 pixels ALONG the filter axis x 8 lines across it, so the 5x1 layer is the 1x5 layer with the
//     roles of x and y exchanged: only the strides handed to the address computations differ (the DMA's per-lane
//     source addresses do the "transpose" for free), and the halo is 4 records per line for both;
//   * the weights of the workgroup's 64 couts (5 taps x 128 cin = 80 KiB) stay resident, which leaves 72 KiB for the
//     activations.  A whole 64-channel chunk per buffer (2 x 36 KiB, one chunk in flight) ran at ~2 x HBM latency per
//     tile (10 us vs ~3 us of MFMA work), so K is streamed in FOUR 32-channel units per tile (64-byte records,
//     18 KiB) through a 4-slot ring, handed over in PAIRS (round 3: two barriers per tile; the pair after the one on the
//     matrix cores is in flight.  Rounds 1-2: one barrier per unit, three units in flight, counted s_waitcnt vmcnt);
//   * epilogues of the GRU: sigmoid (z), sigmoid * h (r*h), (1-z)*h + z*tanh(.) (state update); h and z are
//     prefetched into registers before the MFMA phases.
// ======================================================================================================
constexpr int S_LL = TW + 4;                                    // records per line: 32 + 2x2 halo
constexpr int S_NI = TH * S_LL / 16;                            // 18 DMA instructions per unit (16 records x 64 B each)
constexpr int S_BUF_BYTES = S_NI * 1024;                        // 18,432 B
constexpr int S_NBUF = 4;
constexpr int S_WBYTES = 2 * 5 * 4 * 2 * 1024;                  // chunks x taps x k-steps x cout subtiles x 1 KiB
constexpr int S_LDS_BYTES = S_WBYTES + S_NBUF * S_BUF_BYTES + 1024;   // + bias
static_assert(TH * S_LL % 16 == 0, "unit buffer must be a whole number of DMA instructions");
static_assert(3 * S_NI <= 63, "three units in flight must be countable in vmcnt");
enum { SEP_SIG = 0, SEP_MUL = 1, SEP_GRU = 2 };
#ifndef DEMFI_SEP_GRU_PREFETCH_UNIT
#define DEMFI_SEP_GRU_PREFETCH_UNIT 1
#endif
constexpr int SEP_GRU_PREFETCH_UNIT = DEMFI_SEP_GRU_PREFETCH_UNIT;
#ifndef DEMFI_SEP_PAIRS
#define DEMFI_SEP_PAIRS 1        // round 3: -5 % (z|r) / -6 % (q) against one barrier per unit, same box
#endif
constexpr bool SEP_PAIRS = DEMFI_SEP_PAIRS != 0;

struct SepArgs {
    int t_first, t_end, t_step, nh_shift, cb;
    int tiles_l, tiles_img, Llen, Slen;
    bool tr;
};

__device__ __forceinline__ void sep_item_coords(const SepArgs& a, int it, int& bimg, int& os0, int& ol0)
{
    const int t = it >> a.nh_shift;
    bimg = t / a.tiles_img;
    const int rem = t - bimg * a.tiles_img;
    const int ts = rem / a.tiles_l;
    os0 = ts * TH;
    ol0 = (rem - ts * a.tiles_l) * TW;
}

template <int EPI, int VAR>   // VAR (ablation builds only): 0 product, 1 no epilogue, 2 no MFMA phase, 3 no unit DMA, 4 no aux prefetch
__device__ __forceinline__ void sep_mfma_waves(const demfi_conv* __restrict__ d, const SepArgs& a, const char* wlds,
                                               const char* tbuf, const float* bias_lds, int wave, int lane)
{
    constexpr int NCO = 2;
    const int hi = lane >> 5, lx = lane & 31;
    const demfi_seg& sg = d->segs[d->sub_seg[a.cb * 2]];
    half_t* const dstp = (half_t*)sg.dst.ptr;
    const half_t* const resp = (const half_t*)sg.res.ptr;
    const half_t* const auxp = (const half_t*)sg.aux.ptr;
    // strides along the filter axis (l) and across it (s)
    const int64_t d_sl = a.tr ? sg.dst.sy : sg.dst.sx, d_ss = a.tr ? sg.dst.sx : sg.dst.sy, d_sb = sg.dst.sb;
    const int64_t r_sl = a.tr ? sg.res.sy : sg.res.sx, r_ss = a.tr ? sg.res.sx : sg.res.sy, r_sb = sg.res.sb;
    const int64_t z_sl = a.tr ? sg.aux.sy : sg.aux.sx, z_ss = a.tr ? sg.aux.sx : sg.aux.sy, z_sb = sg.aux.sb;
    const int ch0 = d->oct_ch[a.cb * 8];
    const int Llen = a.Llen, Slen = a.Slen;
    int boff[10];                                               // [tap*2 + k]: 64-byte record (lx + tap) + swizzled 16-byte slot
#pragma unroll
    for (int g = 0; g < 10; ++g) {
        const int col = lx + (g >> 1);
        boff[g] = col * 64 + ((((g & 1) * 2 + hi) ^ ((col >> 2) & 3)) << 4);
    }
    const char* const wl = wlds + lane * 16;
    int ub = 0;                                                 // ring slot of the next unit
    [[maybe_unused]] int trk = -1;
    for (int it = a.t_first; it < a.t_end; it += a.t_step) {
        ++trk;
        int bimg, os0, ol0;
        sep_item_coords(a, it, bimg, os0, ol0);
        // h (and z) of this tile: unconditional clamped loads issued before the MFMA phases (see the 3x3 kernel)
        u4_t rreg[NCO][2][2] = {}, zreg[NCO][2][2] = {};
        auto prefetch_aux = [&]() {
          if constexpr (EPI != SEP_SIG && VAR != 4 && VAR != 1) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int os = min(os0 + wave * 2 + p, Slen - 1), ol = min(ol0 + lx, Llen - 1);
                const half_t* rp = resp + bimg * r_sb + os * r_ss + ol * r_sl + ch0 + hi * 8;
#pragma unroll
                for (int s = 0; s < NCO; ++s) {
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) rreg[s][p][m2] = *gcp<u4_t>(rp + s * 32 + m2 * 16);
                }
                if constexpr (EPI == SEP_GRU) {
                    const half_t* zp = auxp + bimg * z_sb + os * z_ss + ol * z_sl + ch0 + hi * 8;
#pragma unroll
                    for (int s = 0; s < NCO; ++s) {
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) zreg[s][p][m2] = *gcp<u4_t>(zp + s * 32 + m2 * 16);
                    }
                }
            }
          }
        };
        // GRU update (16 loads): issued inside the MFMA phase, behind the second unit's barrier -- before the phase they queue
        // behind the previous tile's stores in the CU's memory pipe and the wave spends ~3 000 cycles issuing them
        // (profiles/r03_phase_trace_gru.txt); r * h (8 loads, no stall measured): before the phase as ever
        if constexpr (EPI != SEP_GRU) prefetch_aux();
        f16x_t acc[NCO][2];
#pragma unroll
        for (int s = 0; s < NCO; ++s) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[s][0][i] = 0.0f; acc[s][1][i] = 0.0f; }
        }
        if constexpr (SEP_PAIRS) {
            // ---- the units in pairs: one barrier, then the ten taps of the two units as ONE software pipeline (the per-unit version
            //      restarted it -- two exposed LDS round trips -- at every unit)
            static_for<0, 2>([&](auto P_) {
                constexpr int pr = decltype(P_)::value;
                if constexpr (pr == 0) TRACE_STAMP(wave, trk, 0);
                if constexpr (pr == 1) TRACE_STAMP(wave, trk, 4);
                asm volatile("s_barrier" ::: "memory");
                if constexpr (pr == 0) TRACE_STAMP(wave, trk, 1);
                if constexpr (pr == 1) TRACE_STAMP(wave, trk, 5);
                const char* const tb0 = tbuf + ub * S_BUF_BYTES + (wave * 2) * (S_LL * 64);
                const char* const tb1 = tbuf + ((ub + 1) & (S_NBUF - 1)) * S_BUF_BYTES + (wave * 2) * (S_LL * 64);
                ub = (ub + 2) & (S_NBUF - 1);
                if constexpr (EPI == SEP_GRU && pr == (SEP_GRU_PREFETCH_UNIT >> 1)) {
                    prefetch_aux();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (VAR == 2) return;
                auto load_tap = [&](FragSet<NCO>& f, auto T_) {       // T = 5 * (unit of the pair) + tap
                    constexpr int T = decltype(T_)::value, q = 2 * pr + T / 5, tap = T % 5;
                    const char* const tb = T < 5 ? tb0 : tb1;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
#pragma unroll
                        for (int s = 0; s < NCO; ++s)
                            f.a[k][s] = *(const uint4*)(wl + ((((q >> 1) * 5 + tap) * 4 + (q & 1) * 2 + k) * NCO + s) * 1024);
                        const char* p0 = tb + boff[tap * 2 + k];
                        f.b[k][0] = *(const uint4*)(p0);
                        f.b[k][1] = *(const uint4*)(p0 + S_LL * 64);
                    }
                };
                auto mma_tap = [&](const FragSet<NCO>& f) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
#pragma unroll
                        for (int s = 0; s < NCO; ++s) {
                            Mma<half_t>::run(acc[s][0], f.a[k][s], f.b[k][0]);
                            Mma<half_t>::run(acc[s][1], f.a[k][s], f.b[k][1]);
                        }
                    }
                };
                auto interleave = [&]() {                               // the 8 ds_reads of the next tap 1:1 with the 8 MFMAs of the current one
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                FragSet<NCO> f[2];
                load_tap(f[0], std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                load_tap(f[1], std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                mma_tap(f[0]);
                __builtin_amdgcn_sched_barrier(0);
                static_for<2, 10>([&](auto T_) {
                    constexpr int T = decltype(T_)::value;
                    load_tap(f[T & 1], T_);
                    mma_tap(f[(T - 1) & 1]);
                    interleave();
                });
                mma_tap(f[1]);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
        static_for<0, 4>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;              // unit q: channels 32q .. 32q+31 of the 128
            // unit q of this tile is in ring slot ub (the DMA wave waited for it); raw barrier: nothing of this wave
            // has to drain (its stores and aux loads stay in flight)
            if constexpr (q == 0) TRACE_STAMP(wave, trk, 0);
            if constexpr (q == 2) TRACE_STAMP(wave, trk, 4);    // arrival at the third unit's barrier
            if constexpr (!SEP_PAIRS || (q & 1) == 0) asm volatile("s_barrier" ::: "memory");
            if constexpr (q == 0) TRACE_STAMP(wave, trk, 1);
            if constexpr (q == 2) TRACE_STAMP(wave, trk, 5);
            const char* tb = tbuf + ub * S_BUF_BYTES + (wave * 2) * (S_LL * 64);
            ub = (ub + 1) & (S_NBUF - 1);
            if constexpr (EPI == SEP_GRU && q == SEP_GRU_PREFETCH_UNIT) {
                prefetch_aux();
                __builtin_amdgcn_sched_barrier(0);
            }
            auto load_tap = [&](FragSet<NCO>& f, int tap) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    // packed weights: [chunk q/2][tap][ks = (q&1)*2 + k][subtile]
#pragma unroll
                    for (int s = 0; s < NCO; ++s)
                        f.a[k][s] = *(const uint4*)(wl + ((((q >> 1) * 5 + tap) * 4 + (q & 1) * 2 + k) * NCO + s) * 1024);
                    const char* p0 = tb + boff[tap * 2 + k];
                    f.b[k][0] = *(const uint4*)(p0);
                    f.b[k][1] = *(const uint4*)(p0 + S_LL * 64);
                }
            };
            auto mma_tap = [&](const FragSet<NCO>& f) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
#pragma unroll
                    for (int s = 0; s < NCO; ++s) {
                        Mma<half_t>::run(acc[s][0], f.a[k][s], f.b[k][0]);
                        Mma<half_t>::run(acc[s][1], f.a[k][s], f.b[k][1]);
                    }
                }
            };
            if constexpr (VAR == 2) return;
            // the 8 ds_reads of the next tap are interleaved 1:1 with the 8 MFMAs of the current one
            auto interleave = [&]() {
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            FragSet<NCO> f0, f1;
            load_tap(f0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_tap(f1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(f0);
            __builtin_amdgcn_sched_barrier(0);
            load_tap(f0, 2);
            mma_tap(f1);
            interleave();
            load_tap(f1, 3);
            mma_tap(f0);
            interleave();
            load_tap(f0, 4);
            mma_tap(f1);
            interleave();
            mma_tap(f0);
            __builtin_amdgcn_sched_barrier(0);
        });
        }
#if defined(DEMFI_TRACE) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int s = 0; s < NCO; ++s) { asm volatile("" ::"v"(acc[s][0])); asm volatile("" ::"v"(acc[s][1])); }
        TRACE_STAMP(wave, trk, 2);
#endif
        // ---- register epilogue ----
        if constexpr (VAR == 1) {
#pragma unroll
            for (int s = 0; s < NCO; ++s) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::"v"(acc[s][0]));
                asm volatile("" ::"v"(acc[s][1]));
#endif
            }
            continue;
        }
        if constexpr (EPI != SEP_SIG) {
#pragma unroll
            for (int s = 0; s < NCO; ++s) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    asm volatile("" : "+v"(rreg[s][q >> 1][q & 1]));
                    if constexpr (EPI == SEP_GRU) asm volatile("" : "+v"(zreg[s][q >> 1][q & 1]));
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NCO; ++s) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const f4_t b0 = *(const f4_t*)(bias_lds + s * 32 + (2 * m2) * 8 + hi * 4);       // bias in MFMA-row order
                const f4_t b1 = *(const f4_t*)(bias_lds + s * 32 + (2 * m2 + 1) * 8 + hi * 4);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    // the bias in LDS is pre-multiplied by the exponent's scale K (see the kernel): 2^(K acc + K b) is one fma + v_exp_f32
                    constexpr float K = EPI == SEP_GRU ? 2.8853900817779268f : -1.4426950408889634f;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {              // cout_perm: quads 2*m2, 2*m2+1 of this lane = channels 16*m2 + 8*hi + 0..7
                        v[j] = __builtin_fmaf(acc[s][p][(2 * m2) * 4 + j], K, b0[j]);
                        v[4 + j] = __builtin_fmaf(acc[s][p][(2 * m2 + 1) * 4 + j], K, b1[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[j]));   // sigmoid(x) resp. 1 / (1 + e^(2x))
                    if constexpr (EPI == SEP_MUL) {
                        const h8_t r = __builtin_bit_cast(h8_t, rreg[s][p][m2]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], (float)r[j], 0.0f);     // one v_fma_mix{lo,hi}_f16: product and fp16 rounding
                    } else if constexpr (EPI == SEP_GRU) {
                        const u4_t rr = rreg[s][p][m2];
                        const h8_t r = __builtin_bit_cast(h8_t, rr);
                        const h8_t z = __builtin_bit_cast(h8_t, zreg[s][p][m2]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            // (1 - z) h + z tanh(x) = h + z (tanh(x) - h), tanh(x) = 1 - 2 / (1 + e^(2x)): fma, v_fma_mix_f32 (tanh - h, h read
                            // as fp16), v_fma_mix_f16 (z, h as fp16; fp16 result) -- 7 VALU per element with the exponent's fma; fp16 path only
                            const float q = __builtin_fmaf(-2.0f, v[j], 1.0f);
                            const float dlt = (j & 1) ? sub_mix_hi(q, rr[j >> 1]) : sub_mix_lo(q, rr[j >> 1]);
                            v[j] = __builtin_fmaf((float)z[j], dlt, (float)r[j]);
                        }
                    }
                    const int os = os0 + wave * 2 + p, ol = ol0 + lx;
                    if (os < Slen && ol < Llen)
                        store8<half_t>(dstp + bimg * d_sb + os * d_ss + ol * d_sl + ch0 + s * 32 + m2 * 16 + hi * 8, v);
                }
            }
        }
        TRACE_STAMP(wave, trk, 3);
    }
}

#ifndef DEMFI_S_NDMA
#define DEMFI_S_NDMA 2
#endif
constexpr int S_NDMA = DEMFI_S_NDMA;                            // waves issuing the unit DMA (S_NI must divide evenly: exact vmcnt counts)
static_assert(S_NI % S_NDMA == 0, "unit DMA instructions must split evenly over the DMA waves");
template <int VAR>
__global__ __launch_bounds__(NT + 64 * S_NDMA, 1) void conv_sep5_c128_persist_kernel(const demfi_conv* __restrict__ d)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    SepArgs a;
    a.tr = d->kh == 5;
    a.Llen = a.tr ? d->H : d->W;
    a.Slen = a.tr ? d->W : d->H;
    a.tiles_l = (a.Llen + TW - 1) / TW;
    a.tiles_img = a.tiles_l * ((a.Slen + TH - 1) / TH);
    a.nh_shift = d->cout_pad == 128 ? 1 : 0;
    const int total = (a.tiles_img * d->batch) << a.nh_shift;
    char* const wlds = smem;
    char* const tbuf = smem + S_WBYTES;
    // work items (tile, cout half), half = item & 1 for the 128-cout launch.  Every stride below is even, so a
    // workgroup keeps ONE half (= one resident weight set) for its whole sequence.
    const int G = gridDim.x;
    if ((G & 15) == 0 && total >= G) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = (((total + 7) >> 3) + 1) & ~1, lo = xcd * q;          // band size rounded up to even: even band starts
        a.t_first = lo + idx;
        a.t_end = min(lo + q, total);
        a.t_step = G >> 3;
    } else {
        a.t_first = blockIdx.x;
        a.t_end = total;
        a.t_step = G;
    }
    if (a.t_first >= a.t_end) return;                           // uniform per workgroup
    a.cb = a.t_first & ((1 << a.nh_shift) - 1);

    if (wave >= 4) {
        // ================= DMA waves (instruction i of a unit belongs to wave i % S_NDMA) ======================
        if (DEMFI_KNOB_BIT(1)) __builtin_amdgcn_s_setprio(3);
        const int dw = wave - 4;
        const demfi_piece& p0 = d->pieces[d->chunks[0].first_piece];
        const demfi_piece& p1 = d->pieces[d->chunks[1].first_piece];
        const char* const src0 = (const char*)p0.v.ptr;
        const char* const src1 = (const char*)p1.v.ptr;
        const int64_t s_l = (a.tr ? p0.v.sy : p0.v.sx) * 2, s_s = (a.tr ? p0.v.sx : p0.v.sy) * 2, sb = p0.v.sb * 2;   // bytes
        const int64_t s_l1 = (a.tr ? p1.v.sy : p1.v.sx) * 2, s_s1 = (a.tr ? p1.v.sx : p1.v.sy) * 2, sb1 = p1.v.sb * 2;
        const char* const zeros = (const char*)d->zero_page;
        // instruction i covers records 16i..16i+15 (record = line*36 + column, 64 B = 4 slots);
        // lane -> (record 16i + lane/4, physical slot lane%4), logical slot = physical ^ ((column >> 2) & 3)
        int off0[S_NI], off1[S_NI], lc[S_NI];
#pragma unroll
        for (int i = 0; i < S_NI; ++i) {
            const int rec = i * 16 + (lane >> 2);
            const int l = rec / S_LL;
            const int c = rec - l * S_LL;
            const int v = (lane & 3) ^ ((c >> 2) & 3);
            off0[i] = (int)(l * s_s + c * s_l) + v * 16;
            off1[i] = (int)(l * s_s1 + c * s_l1) + v * 16;
            lc[i] = l | (c << 8);
        }
        auto issue_unit = [&](int u) {                          // unit u = (item u/4, 32-channel quarter u%4) -> ring slot u%4
            const int it = a.t_first + (u >> 2) * a.t_step, q = u & 3;
            int bimg, os0, ol0;
            sep_item_coords(a, it, bimg, os0, ol0);
            const bool second = q >= 2;
            const char* base = (second ? src1 + (int64_t)bimg * sb1 + (int64_t)os0 * s_s1 + (int64_t)(ol0 - 2) * s_l1
                                       : src0 + (int64_t)bimg * sb + (int64_t)os0 * s_s + (int64_t)(ol0 - 2) * s_l) + (q & 1) * 64;
            char* dst = tbuf + q * S_BUF_BYTES;
            const bool interior = ol0 >= 2 && ol0 + TW + 2 <= a.Llen && os0 + TH <= a.Slen;
            if (interior) {
#pragma unroll
                for (int i = 0; i < S_NI; ++i) {
                    if ((i % S_NDMA) != dw) continue;            // wave-uniform
                    const char* g = base + (second ? off1[i] : off0[i]);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < S_NI; ++i) {
                    if ((i % S_NDMA) != dw) continue;
                    const int is = os0 + (lc[i] & 255), il = ol0 - 2 + (lc[i] >> 8);
                    const char* g = (is < a.Slen && il >= 0 && il < a.Llen) ? base + (second ? off1[i] : off0[i]) : zeros;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                }
            }
        };
        // resident weights of this workgroup's cout half: LDS [chunk][tap][ks][s] <- packed [chunk][tap][ks][nco subtiles]
        const uint4* wsrc = (const uint4*)d->wpack;
        const int nco = d->nco;
        for (int c = 0; c < 2; ++c) {
            const uint4* wc = wsrc + d->chunks[c].w_off;
            for (int g = dw; g < 20; g += S_NDMA) {
                for (int s = 0; s < 2; ++s)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(wc + (g * nco + a.cb * 2 + s) * 64 + lane),
                        (__attribute__((address_space(3))) void*)(wlds + ((c * 20 + g) * 2 + s) * 1024), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // weights landed: from here on vmcnt counts unit loads only
        const int n_units = 4 * ((a.t_end - a.t_first + a.t_step - 1) / a.t_step);   // >= 4
        if constexpr (SEP_PAIRS) {
            // units handed over in PAIRS: two barriers per tile instead of four; pair k + 1 is issued behind pair k's barrier (the MFMA waves
            // have finished pair k - 1 when they arrive there) and has one pair's MFMA time to land
            issue_unit(0);
            issue_unit(1);
            issue_unit(2);
            issue_unit(3);
            const int n_pairs = n_units >> 1;
            for (int k = 0; k < n_pairs; ++k) {
                if (k == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * S_NI / S_NDMA) : "memory");
                else        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((k & 1) == 0) TRACE_STAMP(wave, k >> 1, 0);
                __syncthreads();
                if ((k & 1) == 0) TRACE_STAMP(wave, k >> 1, 1);
                if (k >= 1 && k + 1 < n_pairs) { issue_unit(2 * k + 2); issue_unit(2 * k + 3); }
                if ((k & 1) == 0) TRACE_STAMP(wave, k >> 1, 2);
            }
            return;
        }
        issue_unit(0);
        issue_unit(1);
        issue_unit(2);
        for (int u = 0; u < n_units; ++u) {
            // units u+1, u+2 (if they exist) may stay in flight; loads retire in order
            if (u + 2 < n_units)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * S_NI / S_NDMA) : "memory");
            else if (u + 1 < n_units) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S_NI / S_NDMA) : "memory");
            else                      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((u & 3) == 0) TRACE_STAMP(wave, u >> 2, 0);
            __syncthreads();                                    // hand unit u to the MFMA waves
            if ((u & 3) == 0) TRACE_STAMP(wave, u >> 2, 1);
            // ring slot of unit u+3 = slot of unit u-1: every MFMA wave finished reading it before reaching this barrier
            if (VAR != 3 && u + 3 < n_units) issue_unit(u + 3);
            if ((u & 3) == 0) TRACE_STAMP(wave, u >> 2, 2);
        }
        return;
    }

    // ================= MFMA waves ============================================================================
    float* const bias_lds = (float*)(tbuf + S_NBUF * S_BUF_BYTES);
    const demfi_seg& sg = d->segs[d->sub_seg[a.cb * 2]];
    // bias pre-multiplied by the scale of the epilogue's exponent: e^(2x) = 2^(2 log2(e) x) (tanh), e^(-x) = 2^(-log2(e) x) (sigmoid)
    if (tid < 64) bias_lds[tid] = d->bias[a.cb * 64 + tid] * (sg.mode == DEMFI_MODE_GRU ? 2.8853900817779268f : -1.4426950408889634f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (sg.mode == DEMFI_MODE_GRU)      sep_mfma_waves<SEP_GRU, VAR>(d, a, wlds, tbuf, bias_lds, wave, lane);
    else if (sg.mode == DEMFI_MODE_MUL) sep_mfma_waves<SEP_MUL, VAR>(d, a, wlds, tbuf, bias_lds, wave, lane);
    else                                sep_mfma_waves<SEP_SIG, VAR>(d, a, wlds, tbuf, bias_lds, wave, lane);
}

static bool sep_eligible(const demfi_conv* h)
{
    if (h->dtype != DEMFI_F16 || h->stride != 1 || h->zero_page == nullptr) return false;
    if (!((h->kh == 1 && h->kw == 5) || (h->kh == 5 && h->kw == 1))) return false;
    if (h->pad_y != h->kh / 2 || h->pad_x != h->kw / 2 || h->inH != h->H || h->inW != h->W) return false;
    if (h->n_chunks != 2 || !((h->cout_pad == 64 && h->nco == 2) || (h->cout_pad == 128 && h->nco == 4))) return false;
    for (int c = 0; c < 2; ++c) {
        const demfi_chunk& ch = h->chunks[c];
        if (ch.n_pieces != 1 || ch.nks != 4) return false;
        const demfi_piece& p = h->pieces[ch.first_piece];
        if (!p.fat || p.nch != 64 || p.up_shift || p.v.ptr == nullptr || p.v.sc != 1 || p.v.is_f32) return false;
        // 32-bit per-lane offsets inside a tile
        if (p.v.sy * 2 * 40 >= (int64_t)1 << 31 || p.v.sx * 2 * 40 >= (int64_t)1 << 31) return false;
    }
    for (int cb = 0; cb < h->cout_pad / 64; ++cb) {
        const int sgi = h->sub_seg[cb * 2];
        if (sgi < 0 || h->sub_seg[cb * 2 + 1] != sgi) return false;
        for (int o = 0; o < 8; ++o)
            if (h->oct_seg[cb * 8 + o] != sgi || h->oct_n[cb * 8 + o] != 8 || h->oct_ch[cb * 8 + o] != h->oct_ch[cb * 8] + 8 * o)
                return false;
        const demfi_seg& sg = h->segs[sgi];
        if (sg.scale != 1 || sg.dy || sg.dx || sg.dst.is_f32 || sg.dst.sc != 1) return false;
        if (sg.mode == DEMFI_MODE_STORE) {
            if (sg.act != DEMFI_ACT_SIGMOID || sg.res.ptr != nullptr) return false;
        } else if (sg.mode == DEMFI_MODE_MUL) {
            if (sg.res.ptr == nullptr || sg.res.is_f32 || sg.res.sc != 1) return false;
        } else if (sg.mode == DEMFI_MODE_GRU) {
            if (sg.res.ptr == nullptr || sg.aux.ptr == nullptr || sg.res.is_f32 || sg.aux.is_f32 || sg.res.sc != 1 || sg.aux.sc != 1)
                return false;
        } else {
            return false;
        }
    }
    return true;
}

template <int VAR = 0>
static int launch_sep(const demfi_conv* h, const demfi_conv* dev, hipStream_t st)
{
    DEMFI_LDS_ATTR((conv_sep5_c128_persist_kernel<VAR>));
    const bool tr = h->kh == 5;
    const int Llen = tr ? h->H : h->W, Slen = tr ? h->W : h->H;
    const int total = ((Llen + TW - 1) / TW) * ((Slen + TH - 1) / TH) * h->batch * (h->cout_pad / 64);
    const int grid = total >= 256 ? 256 : total;              // total < 256: one item per workgroup (stride = total, even for 2 halves)
    hipLaunchKernelGGL(conv_sep5_c128_persist_kernel<VAR>, dim3(grid), dim3(NT + 64 * S_NDMA), (size_t)S_LDS_BYTES, st, dev);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}


// ======================================================================================================
// Streamed-weight kernel for the wide-K layer of the network: Ch_Reducer, 7x7, 3 x 64 -> 64 channels (DeMFInet.py:37, 114:
// 9 408 multiply-adds per output value, 1.2 MB of weights -- nothing of it can stay resident in LDS).  Round 3.
//   * one workgroup of FOUR waves per CU (one wave per SIMD: 512 registers each), persistent over 16 x 32-pixel output tiles;
//     wave w = cout half (w & 1) x row half (w >> 1): EIGHT 32x32 accumulators (8 rows x 32 pixels x 32 couts) per wave;
//   * K is walked in units of 32 input channels (the descriptor's chunks: rec_bytes = 64).  A unit's haloed tile (22 lines x 40 records of 64 bytes, XOR-swizzled
//     slots, 55 KiB) is fetched by LDS-DMA into a double buffer while the previous unit is on the matrix cores -- by the MFMA
//     waves themselves, one instruction every six steps, so the loads sit in the wave's ordinary in-order vmcnt stream;
//   * inside a unit the steps are (kx, k-step, ky) with ky innermost: the B fragment of input line r serves output row p at
//     ky = r - p, so a step needs ONE new ds_read_b128 for its 8 MFMAs (a rolling window of 8 lines);
//   * the A fragment (32 couts x 16 channels of one tap) of a step is ONE global_load_dwordx4 straight from the packed
//     weights (L2-resident: 1.2 MB, read by every workgroup in the same order), prefetched 14 steps = 2 groups ahead into a
//     register ring: no weight traffic through LDS, no barrier inside a unit (the general kernel: one per tap, 147 per tile).
//     One raw s_barrier per unit (784 MFMAs = 25 000 matrix-pipe cycles per wave).
//   => per step: 8 MFMAs, 1-2 ds_read_b128, 1 global_load_dwordx4.
// ======================================================================================================
// Round 5: the same kernel instantiated for the RDB growth convolutions of FF_RDB (3x3, 96 + 32 k -> 32 channels at half resolution,
// DeMFInet.py:266-281: 48 launches per window that the general kernel ran at 0.12-0.16 of the matrix peak, per-tile-bound on its
// gather + nine per-tap barriers): NCH = 1 cout half, the four waves are four row groups of a 32 x 32-pixel tile (eight accumulators
// each, so still one A load per 8 MFMAs), units of 32 channels may come from pieces with different strides (the block input and the
// 128-channel growth buffer), 34 x 34 records per unit with no line padding (two units = 146 KiB of LDS), two DMA instructions per
// step (19 per wave and unit against 18 steps).
#ifndef DEMFI_WS3_DEPTH
#define DEMFI_WS3_DEPTH 9
#endif
#ifndef DEMFI_WS_NW
#define DEMFI_WS_NW 4                                            // waves of the streamed-weight kernel's workgroup: 4 (one per SIMD) or 8
#endif
template <int KS, int NW, int NCH = 2, int TH_ = 16> struct WsCfg {
    static constexpr int TH = TH_;                               // output rows of a tile
    static constexpr int RPW = TH / (NW / NCH);                  // output rows (32x32 accumulators) per wave
    static constexpr int BL = KS + RPW - 1;                      // input lines of a wave's rolling B window
    static constexpr int LH = TH + KS - 1;                       // input lines of a tile
    static constexpr int LL = KS == 7 ? ((TW + KS - 1 + 7) & ~7) : TW + KS - 1;   // records per line (TW + KS - 1 used)
    static constexpr int NI = (LH * LL + 15) / 16;               // DMA instructions per unit (16 records x 64 B each)
    static constexpr int UNIT_BYTES = NI * 1024;
    static constexpr int LDS_BYTES = 2 * UNIT_BYTES;
    static constexpr int NG = 2 * KS;                            // (kx, k-step) groups per unit
    static constexpr int NSTEP = NG * KS;
    // A prefetch distance in steps (8 waves: 256 registers per wave).  The 3x3 instantiation runs ONE tile per workgroup on weights no
    // earlier launch has touched: every A fragment is an L2 miss (~2 us) that 240 workgroups take together, so the ring must cover
    // that latency (9 steps of ~190 ns) or the launch is bound by it (depth 6: 35-49 us per layer where ~25 are matrix time)
    static constexpr int DEPTH = NW == 8 ? KS : (KS == 7 ? 2 * KS : DEMFI_WS3_DEPTH);
    static constexpr int NIW = (NI + NW - 1) / NW;               // DMA instructions per wave (the last one may not exist)
    static constexpr int DMA_EVERY = KS == 7 ? 6 : 1;            // DMA instructions are issued every so many steps ...
    static constexpr int DMA_PER = KS == 7 ? 1 : 3;              // ... so many at a time
    static constexpr bool PIECE_STRIDES = KS != 7;               // units may come from pieces with different strides
    static_assert(LDS_BYTES <= 160 * 1024, "two units must fit LDS");
    static_assert((NIW + DMA_PER - 1) / DMA_PER * DMA_EVERY + DEPTH <= NSTEP, "the unit's DMA must be older than the last A fragment consumed in the unit");
    static_assert(NSTEP % DEPTH == 0, "static ring indices");
};

template <int KS, int NW, int NCH = 2, int TH_ = 16>
__global__ __launch_bounds__(64 * NW, 1) void conv_wstream_c64_kernel(const demfi_conv* __restrict__ d)
{
    using C = WsCfg<KS, NW, NCH, TH_>;
    constexpr int WS_TH = C::TH;
    constexpr int RPW = C::RPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lx = lane & 31;
    const int cs = NCH == 2 ? (wave & 1) : 0, rh = NCH == 2 ? (wave >> 1) : wave;      // cout half, row group (RPW rows each)
    const int H = d->H, W = d->W;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + WS_TH - 1) / WS_TH, tiles_img = tiles_x * tiles_y;
    const int total = tiles_img * d->batch;
    int t_first, t_end, t_step;
    {
        const int G = gridDim.x;
        if ((G & 7) == 0 && total >= G) {                       // XCD-aware bands, as the other persistent kernels
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
            const int q = (total + 7) >> 3, lo = xcd * q;
            t_first = lo + idx;
            t_end = min(lo + q, total);
            t_step = G >> 3;
        } else {
            t_first = blockIdx.x;
            t_end = total;
            t_step = G;
        }
    }
    if (t_first >= t_end) return;                               // uniform per workgroup
    const int upt = d->n_chunks;                                // units per tile: the descriptor's 32-channel chunks (rec_bytes = 64)
    const int n_units = ((t_end - t_first + t_step - 1) / t_step) * upt;

    const demfi_piece& p0 = d->pieces[d->chunks[0].first_piece];
    const int64_t sxb = p0.v.sx * 2, syb = p0.v.sy * 2, sbb = p0.v.sb * 2;      // bytes; identical for every piece (eligibility)
    const char* const zeros = (const char*)d->zero_page;
    const uint4* const wbase = (const uint4*)d->wpack;

    // ---- DMA: instruction i = wave + 4 j covers records 16 i .. 16 i + 15; lane -> (record 16 i + lane / 4, physical slot lane % 4),
    //      logical slot (8 channels) = physical ^ ((column >> 2) & 3)
    int doff[C::NIW], dlc[C::NIW];
#pragma unroll
    for (int j = 0; j < C::NIW; ++j) {
        const int rec = min((wave + NW * j) * 16 + (lane >> 2), C::LH * C::LL - 1);      // lanes past the unit (last instruction) re-read its last record
        const int l = rec / C::LL, c = rec - l * C::LL;
        doff[j] = C::PIECE_STRIDES ? (((lane & 3) ^ ((c >> 2) & 3)) << 4) : (int)(l * syb + c * sxb) + (((lane & 3) ^ ((c >> 2) & 3)) << 4);
        dlc[j] = l | (c << 8);
    }
    // ---- B fragments: line (8 rh + r), record (lx + kx), slot (2 ksl + hi) swizzled
    int boff[C::NG];
#pragma unroll
    for (int g = 0; g < C::NG; ++g) {
        const int col = lx + (g >> 1);
        boff[g] = (rh * RPW * C::LL + col) * 64 + ((((g & 1) * 2 + hi) ^ ((col >> 2) & 3)) << 4);
    }

    struct Unit { const char* src; const char* w; int iy0, ix0; bool interior; int sx, sy; };
    const unsigned lane16 = lane * 16;
    auto unit_info = [&](int u) {
        Unit r;
        const int k = u / upt, cu = u - k * upt;
        const int it = t_first + k * t_step;
        const int bimg = it / tiles_img, rem = it - bimg * tiles_img;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        r.iy0 = ty * WS_TH - KS / 2;
        r.ix0 = tx * TW - KS / 2;
        const demfi_piece& pc = d->pieces[d->chunks[cu].first_piece];
        r.sx = C::PIECE_STRIDES ? (int)(pc.v.sx * 2) : (int)sxb;
        r.sy = C::PIECE_STRIDES ? (int)(pc.v.sy * 2) : (int)syb;
        r.src = (const char*)pc.v.ptr + bimg * (C::PIECE_STRIDES ? pc.v.sb * 2 : sbb) + (int64_t)r.iy0 * r.sy + (int64_t)r.ix0 * r.sx;
        r.w = (const char*)(wbase + d->chunks[cu].w_off + cs * 64);                // uniform; + ((tap * 2 + ksl) * NCH) KiB per step, + lane * 16
        r.interior = r.iy0 >= 0 && r.iy0 + C::LH <= H && r.ix0 >= 0 && r.ix0 + C::LL <= W;
        return r;
    };
    auto dma_one = [&](auto J, const Unit& un, char* buf) {
        constexpr int j = decltype(J)::value;
        const int i = wave + NW * j;
        if (i >= C::NI) return;                                  // wave-uniform
        const char* g = un.src + doff[j];
        if constexpr (C::PIECE_STRIDES) g += (dlc[j] & 255) * un.sy + (dlc[j] >> 8) * un.sx;
        if (!un.interior) {
            const int iy = un.iy0 + (dlc[j] & 255), ix = un.ix0 + (dlc[j] >> 8);
            if (!(iy >= 0 && iy < H && ix >= 0 && ix < W)) g = zeros;
        }
        // inline asm, not the builtin: behind the builtin the compiler waits for vmcnt(0) in front of every later ds_read of this
        // wave (the DMA's LDS write may alias it) -- which would drain the A ring 14 times per unit.  The ordering is this kernel's
        // business: nobody reads the buffer before the barrier at the end of the unit.
        const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(buf + i * 1024);
#if defined(__HIP_DEVICE_COMPILE__)
        // m0 (the DMA's LDS base) is a reserved register: naming it as a clobber is undefined behaviour for the compiler (it may keep
        // its own value live across the statement), so the statement saves and restores it -- m0 is unchanged as far as the compiler
        // can tell, and the build treats -Winline-asm as an error so that a clobbered reserved register can never come back.
        unsigned m0_save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_save) : "v"(g), "s"(la) : "memory");
#endif
    };
    auto a_load = [&](const Unit& un, int t) {                  // A fragment of step t = (kx, ksl, ky): SGPR base + lane offset, global
        const int g = t / KS, ky = t - g * KS, kx = g >> 1, ksl = g & 1;
        return __builtin_bit_cast(uint4, *gcp<u4_t>(un.w + (unsigned)((((ky * KS + kx) * 2 + ksl) * NCH) * 1024 + lane16)));
    };

    const demfi_seg& sg = d->segs[d->sub_seg[0]];
    half_t* const dstp = (half_t*)sg.dst.ptr;
    const int act = sg.act;
    const int ch0 = d->oct_ch[0];
    f4_t bq[4];                                                  // bias in MFMA-row order: quads 0..3 of this lane
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) bq[qd] = *gcp<f4_t>(d->bias + cs * 32 + qd * 8 + hi * 4);

    f16x_t acc[RPW];
#pragma unroll
    for (int p = 0; p < RPW; ++p) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[p][i] = 0.0f;
    }
    uint4 A[C::DEPTH];
    // ---- prologue: unit 0 into buffer 0, the first DEPTH A fragments; the DMA is older than the A loads
    Unit cur = unit_info(0);
    static_for<0, C::NIW>([&](auto J) { dma_one(J, cur, smem); });
    static_for<0, C::DEPTH>([&](auto T) { A[decltype(T)::value] = a_load(cur, decltype(T)::value); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::DEPTH) : "memory");
    asm volatile("s_barrier" ::: "memory");

    for (int u = 0; u < n_units; ++u) {
        const bool has_next = u + 1 < n_units;
        TRACE_STAMP(wave, u / upt, u % upt);                    // trace build: start of every unit (6 units per Ch_Reducer tile = the 6 stamp slots)
        const Unit nxt = unit_info(has_next ? u + 1 : 0);
        const char* const tb = smem + (u & 1) * C::UNIT_BYTES;
        char* const nb = smem + ((u + 1) & 1) * C::UNIT_BYTES;
        uint4 B[2][C::BL];
        // the first group's RPW lines (the later groups' are read during the group before)
#pragma unroll
        for (int r = 0; r < RPW; ++r) B[0][r] = *(const uint4*)(tb + boff[0] + r * (C::LL * 64));
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, C::NSTEP>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            constexpr int g = t / KS, ky = t % KS, gb = g & 1;
            const uint4 a = A[t % C::DEPTH];
            // B: this group's next line, and the first RPW lines of the next group (one per step; the rest in the last step)
            if constexpr (ky + RPW < C::BL) B[gb][ky + RPW] = *(const uint4*)(tb + boff[g] + (ky + RPW) * (C::LL * 64));
            if constexpr (g + 1 < C::NG) {
                if constexpr (ky < RPW) B[gb ^ 1][ky] = *(const uint4*)(tb + boff[g + 1] + ky * (C::LL * 64));
                if constexpr (ky == KS - 1) {
#pragma unroll
                    for (int r = KS; r < RPW; ++r) B[gb ^ 1][r] = *(const uint4*)(tb + boff[g + 1] + r * (C::LL * 64));
                }
            }
            // A: the fragment of step t + DEPTH (of the next unit at the end of this one)
            if constexpr (t + C::DEPTH < C::NSTEP) A[t % C::DEPTH] = a_load(cur, t + C::DEPTH);
            else                                   A[t % C::DEPTH] = a_load(nxt, t + C::DEPTH - C::NSTEP);
            // the next unit's tile, one DMA instruction every DMA_EVERY steps
            if constexpr (KS == 7) {
                if constexpr (t % C::DMA_EVERY == 2 && t / C::DMA_EVERY < C::NIW) {
                    if (has_next) dma_one(std::integral_constant<int, t / C::DMA_EVERY>{}, nxt, nb);
                }
            } else {
                if (has_next) {
                    static_for<0, C::DMA_PER>([&](auto Q) {
                        constexpr int j = t * C::DMA_PER + decltype(Q)::value;
                        if constexpr (j < C::NIW) dma_one(std::integral_constant<int, j>{}, nxt, nb);
                    });
                }
            }
#pragma unroll
            for (int p = 0; p < RPW; ++p) Mma<half_t>::run(acc[p], a, B[gb][ky + p]);
            __builtin_amdgcn_sched_barrier(0);
        });
        // every DMA instruction of the next unit is older than the DEPTH A loads still in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::DEPTH) : "memory");
        const int k = u / upt;
        if (u - k * upt == upt - 1) {
            // ---- epilogue of the tile: bias, activation, 16-byte stores (cout_perm: quads 2 m2, 2 m2 + 1 = channels 16 m2 + 8 hi + 0..7)
            const int it = t_first + k * t_step;
            const int bimg = it / tiles_img, rem = it - bimg * tiles_img;
            const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
            const int ox = tx * TW + lx;
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
                const int oy = ty * WS_TH + rh * RPW + p;
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[p][(2 * m2) * 4 + j] + bq[2 * m2][j];
                        v[4 + j] = acc[p][(2 * m2 + 1) * 4 + j] + bq[2 * m2 + 1][j];
                    }
                    apply_act_n<8>(v, act);
                    if (oy < H && ox < W)
                        store8<half_t>(dstp + bimg * sg.dst.sb + oy * sg.dst.sy + ox * sg.dst.sx + ch0 + cs * 32 + m2 * 16 + hi * 8, v);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[p][i] = 0.0f;
            }
        }
        asm volatile("s_barrier" ::: "memory");                 // unit u + 1 landed (every wave waited for its share); buffer u & 1 is free
        cur = nxt;
    }
}

// ks / nch: 7 / 2 = Ch_Reducer (7x7, 64 couts), 3 / 1 = the RDB growth convolutions (3x3, 32 couts, units from pieces of different strides)
static bool wstream_eligible(const demfi_conv* h, int ks = 7, int nch = 2)
{
    if (h->dtype != DEMFI_F16 || h->stride != 1 || h->kh != ks || h->kw != ks || h->pad_y != ks / 2 || h->pad_x != ks / 2) return false;
    if (h->inH != h->H || h->inW != h->W || !h->zero_page || h->rec_bytes != 64) return false;
    if (h->n_chunks < (ks == 7 ? 1 : 2) || h->cout_pad != 32 * nch || h->nco != nch) return false;
    const demfi_piece& p0 = h->pieces[h->chunks[0].first_piece];
    for (int c = 0; c < h->n_chunks; ++c) {
        const demfi_chunk& ch = h->chunks[c];
        if (ch.n_pieces != 1 || ch.nks != 2) return false;
        const demfi_piece& p = h->pieces[ch.first_piece];
        if (!p.fat || p.nch != 32 || p.up_shift || !p.v.ptr || p.v.sc != 1 || p.v.is_f32) return false;
        if (ks == 7 && (p.v.sx != p0.v.sx || p.v.sy != p0.v.sy || p.v.sb != p0.v.sb)) return false;
        if (p.v.sy * 2 * 40 >= (int64_t)1 << 31 || p.v.sx * 2 * 48 >= (int64_t)1 << 31) return false;   // 32-bit per-lane offsets inside a tile
    }
    const int sgi = h->sub_seg[0];
    if (sgi < 0 || (nch == 2 && (h->sub_seg[1] != sgi || h->oct_ch[4] != h->oct_ch[0] + 32))) return false;
    for (int o = 0; o < 4 * nch; ++o)
        if (h->oct_seg[o] != sgi || h->oct_n[o] != 8 || h->oct_ch[o] != h->oct_ch[0] + 8 * o) return false;
    const demfi_seg& sg = h->segs[sgi];
    if (sg.mode != DEMFI_MODE_STORE || sg.scale != 1 || sg.dy || sg.dx || sg.res.ptr || !sg.dst.ptr || sg.dst.is_f32 || sg.dst.sc != 1) return false;
    return true;
}

static bool wstream3_on()
{
    static const bool on = !(getenv("DEMFI_WS3") && atoi(getenv("DEMFI_WS3")) == 0);     // A/B: 0 = the general kernel for the RDB growth convolutions
    return on;
}
static int launch_wstream3(const demfi_conv* h, const demfi_conv* dev, hipStream_t st)
{
    DEMFI_LDS_ATTR((conv_wstream_c64_kernel<3, 4, 1, 32>));
    const int total = ((h->W + TW - 1) / TW) * ((h->H + 31) / 32) * h->batch;
    const int grid = total >= 256 ? 256 : total;
    constexpr size_t lds = WsCfg<3, 4, 1, 32>::LDS_BYTES;
    hipLaunchKernelGGL((conv_wstream_c64_kernel<3, 4, 1, 32>), dim3(grid), dim3(256), lds, st, dev);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

static int launch_wstream(const demfi_conv* h, const demfi_conv* dev, hipStream_t st)
{
    DEMFI_LDS_ATTR((conv_wstream_c64_kernel<7, DEMFI_WS_NW>));
    const int total = ((h->W + TW - 1) / TW) * ((h->H + 15) / 16) * h->batch;
    const int grid = total >= 256 ? 256 : total;
    constexpr size_t lds = WsCfg<7, DEMFI_WS_NW>::LDS_BYTES;
    hipLaunchKernelGGL((conv_wstream_c64_kernel<7, DEMFI_WS_NW>), dim3(grid), dim3(64 * DEMFI_WS_NW), lds, st, dev);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}


// epilogue of the persistent 3x3 kernels: ONE NHWC fp16 destination holding all NCO*32 channels (optional residual)
static bool persist_out_eligible(const demfi_conv* h, bool allow_tanh = false);

bool persist_eligible(const demfi_conv* h)
{
    if (h->dtype != DEMFI_F16 || h->stride != 1 || h->kh != 3 || h->kw != 3 || h->pad_y != 1 || h->pad_x != 1) return false;
    if (h->n_chunks != 1 || h->n_pieces != 1 || h->chunks[0].nks != 4 || h->rec_bytes != 128) return false;
    const demfi_piece& p = h->pieces[0];
    if (!p.fat || p.nch != 64 || p.up_shift != 0 || !p.v.ptr || p.v.is_f32) return false;
    return persist_out_eligible(h, h->nco == 2);                  // the staged-store kernel (NCO == 2) also has a tanh epilogue
}

// THIN epilogue of the narrow kernel: <= 32 packed couts, every octet routed to a planar fp32 [C,H,W] destination
// (optional planar fp32 residual), any activation
static bool thin_out_eligible(const demfi_conv* h)
{
    if (h->nco != 1 || h->cout_pad != 32 || !h->zero_page || h->inH != h->H || h->inW != h->W || h->sub_seg[0] >= 0) return false;
    bool any = false;
    for (int g = 0; g < 4; ++g) {
        if (h->oct_n[g] == 0) continue;
        any = true;
        const demfi_seg& sg = h->segs[h->oct_seg[g]];
        if (sg.mode != DEMFI_MODE_STORE || sg.scale != 1 || sg.dy != 0 || sg.dx != 0) return false;
        if (!sg.dst.ptr || !sg.dst.is_f32) return false;            // any element strides (round 3: the parity views of dec3)
        if (sg.res.ptr && !sg.res.is_f32) return false;
    }
    return any;
}

// narrow layers: one chunk of 32 / 64 / 128 bytes built from <= 2 NHWC fp16 pieces + zero padding
static bool narrow_eligible(const demfi_conv* h)
{
    if (h->dtype != DEMFI_F16 || h->stride != 1 || h->kh != h->kw || (h->kh != 3 && h->kh != 7) || h->pad_y != h->kh / 2 || h->pad_x != h->kw / 2)
        return false;
    if (h->n_chunks != 1) return false;
    const demfi_chunk& ch = h->chunks[0];
    if (ch.nks != 1 && ch.nks != 2 && ch.nks != 4) return false;
    if (h->kh == 7 && (ch.nks != 1 || h->nco != 1)) return false;      // 7x7: 16 input channels, 32 outputs (Mixer.conv_delta1)
    int nreal = 0;
    for (int k = 0; k < ch.n_pieces; ++k) {
        const demfi_piece& p = h->pieces[ch.first_piece + k];
        if ((p.lds_ch * 2) % 16 || (p.nch * 2) % 16) return false;
        if (p.v.ptr == nullptr) continue;
        if (!p.fat || p.up_shift != 0 || p.v.is_f32 || p.v.sc != 1) return false;
        if (p.v.sy * 2 * 16 >= (int64_t)1 << 31 || p.v.sx * 2 * 64 >= (int64_t)1 << 31) return false;   // 32-bit offsets inside a tile
        ++nreal;
    }
    if (nreal < 1 || nreal > 2) return false;
    return persist_out_eligible(h) || thin_out_eligible(h);
}

static bool persist_out_eligible(const demfi_conv* h, bool allow_tanh)
{
    if (h->nco > 2 || h->cout_pad != 32 * h->nco || !h->zero_page || h->inH != h->H || h->inW != h->W) return false;
    const int sg = h->sub_seg[0];
    if (sg < 0) return false;
    for (int sb = 0; sb < h->nco; ++sb)
        if (h->sub_seg[sb] != sg || h->oct_ch[sb * 4] != h->oct_ch[0] + 32 * sb) return false;
    const demfi_seg& seg = h->segs[sg];
    if (seg.mode != DEMFI_MODE_STORE || seg.scale != 1 || seg.dy != 0 || seg.dx != 0) return false;
    if (seg.act != DEMFI_ACT_NONE && seg.act != DEMFI_ACT_RELU && !(allow_tanh && seg.act == DEMFI_ACT_TANH)) return false;
    return true;
}

template <typename T, int NCO>
int launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, size_t lds)
{
    const int tiles = ((h->W + TW - 1) / TW) * ((h->H + TH - 1) / TH);
    dim3 grid(tiles, h->cout_pad / (32 * h->nco), h->batch);
    bool all_staged = true;                                      // no subtile needs the direct (thin / planar / ragged) epilogue
    for (int sb = 0; sb < h->cout_pad / 32; ++sb) all_staged = all_staged && h->sub_seg[sb] >= 0;
    static const int nodirect = getenv("DEMFI_CONV_NODIRECT") ? atoi(getenv("DEMFI_CONV_NODIRECT")) : 1;      // A/B switch
    if (all_staged && nodirect) {
        DEMFI_LDS_ATTR((conv_kernel<T, NCO, false>));
        hipLaunchKernelGGL((conv_kernel<T, NCO, false>), grid, dim3(NT), lds, st, dev);
    } else {
        DEMFI_LDS_ATTR((conv_kernel<T, NCO, true>));
        hipLaunchKernelGGL((conv_kernel<T, NCO, true>), grid, dim3(NT), lds, st, dev);
    }
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

// the descriptor belongs to one of the persistent kernels whose epilogue works on 8 consecutive channels per lane: the 64-channel
// 3x3 kernel, the narrow kernel with an NHWC destination, the SepConvGRU kernel.  Their layers are packed with cout_perm.
bool demfi_persist_eligible(const demfi_conv* h)
{
    return sep_eligible(h) || wstream_eligible(h) || (wstream3_on() && wstream_eligible(h, 3, 1)) || persist_eligible(h) || (narrow_eligible(h) && persist_out_eligible(h)) ||
           demfi_ws2_eligible(h);
}

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
#if defined(DEMFI_ABLATION) || defined(DEMFI_TRACE)
    {   // experiment builds only (a synchronous copy on the first launch: never inside a stream capture)
        static const int knob_set = [] {
            const int k = getenv("DEMFI_KNOB") ? atoi(getenv("DEMFI_KNOB")) : 0;
            if (k) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_knob), &k, sizeof(k));
            return k;
        }();
        (void)knob_set;
    }
#endif
    if (h->pack.ptr != nullptr) {
        // the packed copy is an epilogue of the thin-output narrow kernel only: any other layer asking for it must fail loudly
        bool ok = h->dtype == DEMFI_F16 && narrow_eligible(h) && thin_out_eligible(h) && !persist_out_eligible(h) && h->pack.sc == 1 && !h->pack.is_f32;
        for (int g = 0; g < 4; ++g) ok = ok && (h->pack_oct_ch[g] < 0 || (h->pack_oct_ch[g] % 4 == 0 && h->oct_n[g] > 0));
        if (!ok) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d: packed copy (demfi_conv.pack) on a layer that is not a thin-output fp16 narrow layer, or malformed");
    }
    if ((h->cout_perm != 0) != demfi_persist_eligible(h))
        return demfi_set_error(DEMFI_ERR_ARG, h->cout_perm ? "demfi_conv2d: descriptor packed for a persistent kernel (cout_perm) but not eligible for one (zero_page missing?)"
                                                           : "demfi_conv2d: persistent-kernel layer without cout_perm (build the descriptor with demfi_conv_build)");
    if (sep_eligible(h)) {
#ifdef DEMFI_ABLATION
        static const int svar = getenv("DEMFI_SEP_VARIANT") ? atoi(getenv("DEMFI_SEP_VARIANT")) : 0;
        if (svar == 1) return launch_sep<1>(h, dev, st);
        if (svar == 2) return launch_sep<2>(h, dev, st);
        if (svar == 3) return launch_sep<3>(h, dev, st);
        if (svar == 4) return launch_sep<4>(h, dev, st);
        if (svar != -1)
#endif
        return launch_sep(h, dev, st);
    }
    if (wstream_eligible(h)) return launch_wstream(h, dev, st);
    if (!persist_eligible(h) && !narrow_eligible(h) && demfi_ws2_eligible(h)) return demfi_ws2_launch(h, dev, st);      // wsconv.hip (round 6)
    if (wstream3_on() && wstream_eligible(h, 3, 1)) return launch_wstream3(h, dev, st);
    if (persist_eligible(h)) {
#ifdef DEMFI_ABLATION
        static const int var = getenv("DEMFI_PERSIST_VARIANT") ? atoi(getenv("DEMFI_PERSIST_VARIANT")) : 0;
        if (var == -1) goto general;
        if (h->nco == 2 && var == 1) return launch_persist<2, 1>(h, dev, st);
        if (h->nco == 2 && var == 2) return launch_persist<2, 2>(h, dev, st);
        if (h->nco == 2 && var == 3) return launch_persist<2, 3>(h, dev, st);
        if (h->nco == 2 && var == 4) return launch_persist<2, 4>(h, dev, st);
        if (h->nco == 2 && var == 7) return launch_persist<2, 7>(h, dev, st);
        if (h->nco == 2 && var == 9) return launch_persist<2, 9>(h, dev, st);
        if (h->nco == 2 && var == 15) return launch_persist<2, 10>(h, dev, st);
        if (h->nco == 2 && var == 16) return launch_persist<2, 11>(h, dev, st);
#endif
        if (h->nco == 2) {
#ifdef DEMFI_ABLATION
            if (var == 5) return launch_persist<2>(h, dev, st);
            // DEMFI_PAIR: 4 the round-2 product (stores from the MFMA waves); the round-3 double-accumulator experiment (5) was deleted
            // in round 5 (measured negative, profiles/r03_notes.md; git history: conv_exp_dacc.inc)
            static const int pair = getenv("DEMFI_PAIR") ? atoi(getenv("DEMFI_PAIR")) : 0;
            if (pair == 4) return launch_persist<2>(h, dev, st);
#endif
#ifdef DEMFI_TRACE
            if (getenv("DEMFI_PAIR") && atoi(getenv("DEMFI_PAIR")) == 4) return launch_persist<2>(h, dev, st);   // phase trace of the 4-wave kernel
#endif
            return launch_stg(h, dev, st);
        }
        return launch_persist<1>(h, dev, st);
    }
    if (narrow_eligible(h)) {
#ifdef DEMFI_ABLATION
        if (!(getenv("DEMFI_NARROW_OFF") && atoi(getenv("DEMFI_NARROW_OFF"))))
#endif
        {
            const bool thin = !persist_out_eligible(h);
            if (h->kh == 7) {
                if (!thin) return launch_narrow<1, 32, 7>(h, dev, st, false);
            } else
            switch (h->chunks[0].nks * 2 + h->nco) {
            case 1 * 2 + 1: return launch_narrow<1, 32>(h, dev, st, thin);
            case 1 * 2 + 2: return launch_narrow<2, 32>(h, dev, st, thin);
            case 2 * 2 + 1: return launch_narrow<1, 64>(h, dev, st, thin);
            case 2 * 2 + 2: return launch_narrow<2, 64>(h, dev, st, thin);
            case 4 * 2 + 1: return launch_narrow<1, 128>(h, dev, st, thin);
            case 4 * 2 + 2: return launch_narrow<2, 128>(h, dev, st, thin);
            }
        }
    }
#ifdef DEMFI_ABLATION
general:
#endif
    return h->dtype == DEMFI_F16 ? dispatch<half_t>(h, dev, st, (size_t)lds) : dispatch<float>(h, dev, st, (size_t)lds);
}

#ifdef DEMFI_TRACE
// trace build only: copy the phase trace out (see TRACE_STAMP) and clear it
extern "C" int demfi_trace_dump(unsigned long long* out, int64_t n)
{
    const int64_t have = (int64_t)TR_WGS * TR_WAVES * TR_TILES * TR_STAMPS;
    if (!out || n < have) return demfi_set_error(DEMFI_ERR_ARG, "demfi_trace_dump: need room for %lld entries", (long long)have);
    DEMFI_HIP_CHECK(hipDeviceSynchronize());
    DEMFI_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), have * 8));
    static unsigned long long zeros[TR_WGS * TR_WAVES * TR_TILES * TR_STAMPS];
    DEMFI_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), zeros, have * 8));
    return (int)have;
}
#endif
