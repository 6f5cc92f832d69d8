// Shared device helpers of the convolution kernels (conv_general.hip, conv_c64.hip, conv_narrow.hip, conv_sep.hip, conv_wstream.hip):
// tile constants, the MFMA wrappers, the epilogue the general and the 32-cout persistent kernel share, and -- experiment builds only --
// the per-translation-unit knob word / phase-trace buffer (each unit has its own copy; conv.hip collects them).
#pragma once
#include "common.h"
#include "conv_kernels.h"
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
// one k-step's fragments of the 64-channel kernels (conv_c64.hip, conv_sep.hip)
template <int NCO> struct FragSet { uint4 a[2][NCO]; uint4 b[2][2]; };

}  // namespace

// experiment builds: this unit's knob word / trace buffer behind a function conv.hip can call (device globals are per translation unit)
#if defined(DEMFI_ABLATION) || defined(DEMFI_TRACE)
#define DEMFI_TU_KNOB(fn) void fn(int k) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_knob), &k, sizeof(k)); }
#else
#define DEMFI_TU_KNOB(fn) void fn(int) { }
#endif
#ifdef DEMFI_TRACE
#define DEMFI_TU_TRACE(fn)                                                                                  \
    int fn(unsigned long long* acc)                                                                         \
    {                                                                                                       \
        constexpr int64_t have = (int64_t)TR_WGS * TR_WAVES * TR_TILES * TR_STAMPS;                         \
        static unsigned long long tmp[have], zeros[have];                                                   \
        DEMFI_HIP_CHECK(hipMemcpyFromSymbol(tmp, HIP_SYMBOL(g_trace), have * 8));                           \
        for (int64_t i = 0; i < have; ++i) acc[i] |= tmp[i];                                                \
        DEMFI_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), zeros, have * 8));                           \
        return 0;                                                                                           \
    }
#else
#define DEMFI_TU_TRACE(fn) int fn(unsigned long long*) { return 0; }
#endif
