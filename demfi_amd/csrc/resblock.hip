// Fused residual block  y = x + conv2(relu(conv1(x)))  for the 3x3 64 -> 64 fp16 layers of the network (round 5).
//
// Replaces the (conv1, ReLU, conv2, + identity) pairs of ResBlock2D_3D / ResBlock2D (DeMFInet.py:524-563) as they are used by the
// FAC-FB encoder (feature_extraction, 341-344), D1 (Decoder_res, 95-101: Conv3d(1,3,3) == batch 3) and D2 (Decoder_res_2,
// 158-160): 30 of the 36 3x3 64 -> 64 convolutions of a forward.  Layer by layer (conv3x3_c64_stg_kernel) a block moves
// 256 + 384 = 640 B per pixel through HBM and its second half sits on the 2 : 1 read : write memory wall (4.85 TB/s,
// profiles/r04_notes.md section 11); fused, the intermediate never leaves the CU and the identity is taken from the input tile
// that is in LDS anyway: 128 x 1.2 (halo re-reads) + 128 B per pixel.
//
// What makes it fit (profiles/r04_notes.md sections 16-17): the A (weight) fragments do NOT live in LDS.  Like Ch_Reducer's
// streamed-weight kernel every step's A fragment is ONE global_load_dwordx4 from the packed, L2-resident weights into a register
// ring, and a wave owns 8 rows x 32 couts = eight 32x32 accumulators, so one A load and 10/24 B reads feed 8 MFMAs
// (LDS: 0.42 reads per MFMA against 0.83 with resident weights).  The 144 KiB of the two weight sets are out of LDS, which then
// holds: the haloed input window (18 lines x 34 px x 128 B), the intermediate (16 + 2 x 2 carried lines x 32 px) and -- aliased
// with the intermediate -- the staged outputs.
//
// Shape of the walk: a workgroup owns a contiguous run of (image, 30-column strip, 16-row step) items and walks DOWN a strip:
//   step k:  input rows [16k, 16k+17] (DMA by the helper waves, whole window re-fetched: 18/16 of the rows)
//            conv1 -> M rows [16k+1, 16k+16] (32 columns x0-1 .. x0+30, ReLU, zero outside the image), fp16 in LDS
//            conv2 over M rows [16k-1, 16k+16] (the first two carried over from step k-1) -> output rows [16k, 16k+15],
//            accumulators initialised with bias2 + identity (fp32) so the epilogue is a plain convert;
//   a chain (strip start or workgroup start) opens with a conv1-only step on the 16 rows above to produce the carried lines.
// Horizontal halo: a strip computes 32 intermediate columns for 30 outputs (conv1 x 1.07, conv2 32/30 of a tile wide).
// Roles by wave age as in the staged-store kernel: MFMA waves 0-3 (cout half x row half) touch global memory only for their A
// fragments; helper waves 4-7 issue the window DMA (under conv2's MFMA phase), read the staged outputs back 8 lanes per pixel and store
// whole 128-byte lines with the streaming hint (round 6: under conv1's epilogue, not its MFMA phase).  Four raw s_barriers per step of 2 x 288 MFMAs per wave (18 432 matrix-pipe cycles).
#include "common.h"
#include <type_traits>
#include <vector>

namespace {

constexpr int RB_R = 16;                                         // output rows per step
constexpr int RB_OW = 30;                                        // output columns per strip (of the 32-pixel MFMA tile)
constexpr int RB_IW = 34;                                        // input records per window line
constexpr int RB_IL = RB_R + 2;                                  // input lines per window
constexpr int RB_IN_LS = RB_IW * 128;                            // bytes per input line
constexpr int RB_IN_NI = (RB_IL * RB_IW + 7) / 8;                // 77 DMA instructions per window (8 records each)
constexpr int RB_IN_BYTES = RB_IN_NI * 1024;
constexpr int RB_ML = 32 * 128;                                  // bytes per intermediate / staging line
constexpr int RB_M_OFF = RB_IN_BYTES;                            // 16 lines: M rows 2..17 of the window, later the staged outputs
constexpr int RB_C_OFF = RB_M_OFF + RB_R * RB_ML;                // two carry slots of 2 lines (M rows 0, 1 of the window)
constexpr int RB_BIAS_OFF = RB_C_OFF + 2 * 2 * RB_ML + 256;      // + pad: columns 32, 33 of the last line are read (never used)
constexpr int RB_LDS = RB_BIAS_OFF + 2 * 64 * 4;
constexpr int RB_NH = 4;                                         // helper waves
constexpr int RB_NTHREADS = 256 + 64 * RB_NH;
constexpr int RB_NIW = (RB_IN_NI + RB_NH - 1) / RB_NH;           // 20 DMA instructions per helper (the last may not exist)
#ifndef DEMFI_RB_DEPTH
#define DEMFI_RB_DEPTH 4                                         // A prefetch distance in steps of 8 MFMAs (6: 14 registers spilled at the 256-register limit)
#endif
constexpr int RB_DEPTH = DEMFI_RB_DEPTH;
#ifndef DEMFI_RB_STORE_AT
#define DEMFI_RB_STORE_AT 1                                      // round 6: the output stores under conv1's EPILOGUE (0: under its MFMA phase, round 5)
#endif
constexpr int RB_STORE_AT = DEMFI_RB_STORE_AT;
constexpr int RB_NSTEP = 36;                                     // (kx, k-step) groups x ky
static_assert(RB_NSTEP % RB_DEPTH == 0, "static ring indices");
static_assert(RB_LDS <= 160 * 1024, "LDS budget");

// In-kernel phase trace (libdemfi_hip_trace.so only): s_memtime at the barriers of the first RB_TR_STEPS loop iterations of
// workgroups 0..31, [wg][wave][iteration][stamp]; tools/rb_trace.py prints the phase means.
#ifdef DEMFI_TRACE
constexpr int RB_TR_WGS = 32, RB_TR_WAVES = 8, RB_TR_STEPS = 24, RB_TR_STAMPS = 10;
__device__ unsigned long long g_rb_trace[RB_TR_WGS * RB_TR_WAVES * RB_TR_STEPS * RB_TR_STAMPS];
#define RB_STAMP(wave_, k_, i_)                                                                                       \
    do {                                                                                                              \
        if (blockIdx.x < RB_TR_WGS && (k_) < RB_TR_STEPS && (threadIdx.x & 63) == 0)                                  \
            g_rb_trace[((blockIdx.x * RB_TR_WAVES + (wave_)) * RB_TR_STEPS + (k_)) * RB_TR_STAMPS + (i_)] =           \
                __builtin_readcyclecounter() | ((unsigned long long)(pro ? 1 : 0) << 63);                             \
    } while (0)
#else
#define RB_STAMP(wave_, k_, i_) do { } while (0)
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void rb_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rb_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ void rb_mma(f16x_t& acc, const uint4& a, const uint4& b)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void rb_mma_c(f16x_t& acc, const uint4& a, const uint4& b, const f16x_t& c)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
}
// (fp16 half of a packed pair) * 1.0 + c in one VALU op
__device__ __forceinline__ float rb_mix_lo(unsigned a, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}
__device__ __forceinline__ float rb_mix_hi(unsigned a, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}

struct RbArgs {
    const char* src;  int64_t s_sx, s_sy, s_sb;                  // input x (bytes)
    char* dst;        int64_t d_sx, d_sy, d_sb;                  // output y (bytes), already at its first channel
    const char* w1; const char* w2;                              // packed A fragments [tap][k-step][cout half][lane][16 B]
    const float* b1; const float* b2;                            // packed cout order
    const char* zeros;
    int H, W, batch;
    int n_strips, n_rsteps;
};

__global__ __launch_bounds__(RB_NTHREADS, 1) void resblock3x3_c64_kernel(const RbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int per_img = a.n_strips * a.n_rsteps;
    const int total = per_img * a.batch;
    // contiguous run of items per workgroup; the workgroups of an XCD (blockIdx % 8) share a contiguous band, so neighbouring
    // strips (which share 4 of 34 input columns) meet in one L2
    int it0, it1;
    {
        const int G = gridDim.x;
        if ((G & 7) == 0 && total >= G) {
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, nw = G >> 3;
            const int q = total >> 3, r = total & 7;
            const int lo = xcd * q + min(xcd, r), n = q + (xcd < r ? 1 : 0);
            it0 = lo + (int)(((int64_t)n * idx) / nw);
            it1 = lo + (int)(((int64_t)n * (idx + 1)) / nw);
        } else {
            it0 = (int)(((int64_t)total * blockIdx.x) / G);
            it1 = (int)(((int64_t)total * (blockIdx.x + 1)) / G);
        }
    }
    if (it0 >= it1) return;                                      // uniform per workgroup
    // (item, opening?) -> image, first output column, first input row of the window
    auto pos_of = [&](int it, bool pro, int& img, int& x0, int& row0) {
        img = it / per_img;
        const int rem = it - img * per_img;
        const int strip = rem / a.n_rsteps, k = rem - strip * a.n_rsteps;
        x0 = strip * RB_OW;
        row0 = k * RB_R - (pro ? RB_R : 0);
    };
    auto opens_chain = [&](int it) { return (it % a.n_rsteps) == 0; };

    if (wave >= 4) {
        // ================= helper waves: window DMA + the global stores of the staged outputs ============================
        const int dw = wave - 4;
        __builtin_assume(dw >= 0 && dw < RB_NH);
        unsigned off[RB_NIW];
        int lc[RB_NIW];
#pragma unroll
        for (int k = 0; k < RB_NIW; ++k) {
            const int i = dw + RB_NH * k;
            const int rec = i * 8 + (lane >> 3);
            const int rc = min(rec, RB_IL * RB_IW - 1);          // lanes past the window (last instruction) re-read its last record
            const int l = rc / RB_IW, c = rc - l * RB_IW;
            off[k] = (unsigned)(l * a.s_sy + c * a.s_sx) + (((lane & 7) ^ ((c >> 1) & 7)) << 4);
            lc[k] = (i < RB_IN_NI && rec < RB_IL * RB_IW) ? (l | (c << 8)) : 0xffff;
        }
        auto issue_window = [&](int it, bool pro) {
            int img, x0, row0;
            pos_of(it, pro, img, x0, row0);
            const char* base = a.src + (int64_t)img * a.s_sb + (int64_t)row0 * a.s_sy + (int64_t)(x0 - 2) * a.s_sx;
            const bool interior = row0 >= 0 && row0 + RB_IL <= H && x0 - 2 >= 0 && x0 - 2 + RB_IW <= W;
            if (interior) {
#pragma unroll
                for (int k = 0; k < RB_NIW; ++k) {
                    const int i = dw + RB_NH * k;
                    if (i >= RB_IN_NI) continue;                 // wave-uniform
                    __builtin_amdgcn_global_load_lds((const DEMFI_GLOBAL void*)(base + off[k]),
                                                     (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < RB_NIW; ++k) {
                    const int i = dw + RB_NH * k;
                    if (i >= RB_IN_NI) continue;
                    const int iy = row0 + (lc[k] & 255), ix = x0 - 2 + (lc[k] >> 8);
                    const char* g = (lc[k] != 0xffff && iy >= 0 && iy < H && ix >= 0 && ix < W) ? base + off[k] : a.zeros;
                    __builtin_amdgcn_global_load_lds((const DEMFI_GLOBAL void*)g,
                                                     (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, 0, 0);
                }
            }
        };
        // staged outputs: helper dw owns pixels 8 dw .. 8 dw + 7 of every row (lane -> pixel lane / 8, physical slot lane % 8)
        u4_t stage[RB_R];
        const int spx = dw * 8 + (lane >> 3);
        const unsigned dlane = (unsigned)(spx * a.d_sx) + ((((lane & 7) ^ ((spx >> 1) & 7)) * 8) * 2);
        auto stage_read = [&]() {
            const char* sbp = smem + RB_M_OFF + dw * 1024 + lane * 16;
#pragma unroll
            for (int r = 0; r < RB_R; ++r) stage[r] = *(const u4_t*)(sbp + r * RB_ML);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        auto stage_store = [&](int img, int x0, int row0) {
            char* const obase = a.dst + (int64_t)img * a.d_sb + (int64_t)row0 * a.d_sy + (int64_t)x0 * a.d_sx;     // wave-uniform
            const bool px_ok = spx < RB_OW && x0 + spx < W;
            if (row0 + RB_R <= H) {
                if (px_ok) {
#pragma unroll
                    for (int r = 0; r < RB_R; ++r) __builtin_nontemporal_store(stage[r], gp<u4_t>(obase + r * a.d_sy + dlane));
                }
            } else {
#pragma unroll
                for (int r = 0; r < RB_R; ++r)
                    if (px_ok && row0 + r < H) __builtin_nontemporal_store(stage[r], gp<u4_t>(obase + r * a.d_sy + dlane));
            }
        };
        int it = it0;
        bool pro = true, have_prev = false;
        int p_img = 0, p_x0 = 0, p_row0 = 0;
        issue_window(it, pro);
        [[maybe_unused]] int trk = 0;
        for (;;) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the window has landed (and this wave's stores are out)
            RB_STAMP(wave, trk, 0);
            asm volatile("s_barrier" ::: "memory");             // A
            RB_STAMP(wave, trk, 1);
            // DEMFI_RB_STORE_AT: 0 = the 16 output stores right here, under the conv1 MFMA phase (round 5); 1 (product since round 6) =
            // behind barrier B, under conv1's EPILOGUE: the stores and the MFMA waves' A-fragment loads share the CU's one memory pipe, and
            // a matrix phase without the helpers' VMEM traffic runs at 0.93 of the pipe instead of 0.81 (conv1 phase 11 400 -> 9 950
            // cycles, step 28 900 -> 27 400, launch -3 %: profiles/r06_resblock_store_timing_ab.txt); the epilogue issues no VMEM itself
            if (have_prev) { stage_read(); if constexpr (RB_STORE_AT == 0) stage_store(p_img, p_x0, p_row0); }
            RB_STAMP(wave, trk, 2);
            asm volatile("s_barrier" ::: "memory");             // B: the staged outputs are in registers -> the M lines are free
            if constexpr (RB_STORE_AT == 1) { if (have_prev) stage_store(p_img, p_x0, p_row0); }
            RB_STAMP(wave, trk, 3);
            asm volatile("s_barrier" ::: "memory");             // C: every MFMA wave is done with the input window
            RB_STAMP(wave, trk, 4);
            int nit = it;
            bool npro = false, more = true;
            if (!pro) { nit = it + 1; more = nit < it1; npro = more && opens_chain(nit); }
            // (measured negative, round 6: the window DMA as a burst at s_setprio 3 -- issued in 3 300 instead of 8 300 cycles, conv2's MFMA
            // phase 11 900 -> 12 750 cycles: profiles/r06_resblock_store_timing_ab.txt)
            if (more) issue_window(nit, npro);
            RB_STAMP(wave, trk, 5);
            asm volatile("s_barrier" ::: "memory");             // D
            RB_STAMP(wave, trk, 6);
            ++trk;
            have_prev = !pro;
            if (have_prev) pos_of(it, false, p_img, p_x0, p_row0);
            if (!more) break;
            it = nit;
            pro = npro;
        }
        asm volatile("s_barrier" ::: "memory");                 // F: the last step's outputs are staged
        stage_read();
        stage_store(p_img, p_x0, p_row0);
        return;
    }

    // ================= MFMA waves ================================================================================
#ifdef DEMFI_RB_PRIO
    __builtin_amdgcn_s_setprio(DEMFI_RB_PRIO);                  // experiment: MFMA waves above the helper waves of their SIMD
#endif
    const int hi = lane >> 5, lx = lane & 31;
    const int cs = wave & 1, rh = wave >> 1;                    // cout half, row half (8 rows each)
    const unsigned lane16 = lane * 16;
    const char* const w1 = a.w1 + cs * 1024;
    const char* const w2 = a.w2 + cs * 1024;
    // biases to LDS once (packed order == MFMA row order): conv1's as the C operand of every accumulator's first MFMA, conv2's
    // summed with the identity into the accumulators' initial value
    if (rh == 0 && lane < 32) {
        ((float*)(smem + RB_BIAS_OFF))[cs * 32 + lane] = a.b1[cs * 32 + lane];
        ((float*)(smem + RB_BIAS_OFF))[64 + cs * 32 + lane] = a.b2[cs * 32 + lane];
    }
    // carried lines of a strip's first step: M row -1 is outside the image; both slots start as zeros (slot contents are always
    // rewritten by the opening step before they are read, this only keeps uninitialised LDS out of the picture)
    // B fragment offset of (kx, k-step ks): record lx + kx, 16-byte slot (2 ks + hi) XOR-swizzled by the record's column.  The
    // swizzle only touches bits 4-6 of the byte offset and ks only bits 5-6, so the four k-steps of a kx are boff0[kx] ^ (ks << 5):
    // three registers instead of twelve (the kernel lives at the 256-register limit of an 8-wave workgroup)
    int boff0[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = lx + kx;
        boff0[kx] = col * 128 + ((hi ^ ((col >> 1) & 7)) << 4);
    }
    auto boff = [&](auto G) {
        constexpr int g = decltype(G)::value;
        int b = boff0[g >> 2];
        asm volatile("" : "+v"(b));                              // computed where it is used (1 VALU): hoisted out of the step loops the twelve values are spilled
        return b ^ ((g & 3) << 5);
    };
    int soff[2], ioff[2];
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) {
        soff[m2] = lx * 128 + (((cs * 4 + m2 * 2 + hi) ^ ((lx >> 1) & 7)) << 4);
        ioff[m2] = (lx + 2) * 128 + (((cs * 4 + m2 * 2 + hi) ^ (((lx + 2) >> 1) & 7)) << 4);
    }
    auto a_load = [&](const char* w, auto T) {
        constexpr int t = decltype(T)::value;
        constexpr int g = t / 3, ky = t - 3 * g, kx = g >> 2, ks = g & 3;
        // uniform base (SGPR pair, 2 SALU per load) + the lane's 32-bit offset: the saddr form.  Left to itself the compiler hoists 72
        // per-lane 64-bit addresses out of the step loop and spills them.
        const char* wb = w;
        asm volatile("" : "+s"(wb));                             // opaque uniform base + (constant + lane offset) as one 32-bit VGPR: the saddr form, 1 VALU per load, nothing hoisted
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));
        return __builtin_bit_cast(uint4, *gcp<u4_t>(wb + (unsigned)((((ky * 3 + kx) * 4 + ks) * 2) * 1024 + l16)));
    };
    uint4 A[RB_DEPTH];
    f16x_t acc[8];
    rb_for<0, RB_DEPTH>([&](auto T) { A[decltype(T)::value] = a_load(w1, T); });

    // one convolution phase: 36 steps of 8 MFMAs.  line(L, o): address of window line L (0..9, relative to this wave's first
    // line) + fragment offset o.  The A ring runs through the phase boundary (wnxt = the next phase's weights).
    auto conv_phase = [&](auto line, const char* wcur, const char* wnxt, auto INITC, const f16x_t& cinit) {
        // B window: a step (g, ky) multiplies output row p with window line ky + p.  Lines 8, 9 of a group arrive during its steps
        // 0, 1; the NEXT group's lines 0..7 are read during step 2, each right behind the MFMA that uses line 2 + p for the last
        // time, into the register that just died: 9-10 fragments live instead of 16.
        uint4 B[10];
        {
            const int o = boff(std::integral_constant<int, 0>{});
            rb_for<0, 8>([&](auto R) { B[decltype(R)::value] = *(const uint4*)line(R, o); });
        }
        __builtin_amdgcn_sched_barrier(0);
        rb_for<0, RB_NSTEP>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            constexpr int g = t / 3, ky = t % 3;
            const uint4 av = A[t % RB_DEPTH];
            if constexpr (ky < 2) B[ky + 8] = *(const uint4*)line(std::integral_constant<int, ky + 8>{}, boff(std::integral_constant<int, g>{}));
            if constexpr (t + RB_DEPTH < RB_NSTEP) A[t % RB_DEPTH] = a_load(wcur, std::integral_constant<int, t + RB_DEPTH>{});
            else                                   A[t % RB_DEPTH] = a_load(wnxt, std::integral_constant<int, t + RB_DEPTH - RB_NSTEP>{});
            if constexpr (ky < 2 || g + 1 == 12) {
                rb_for<0, 8>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    if constexpr (decltype(INITC)::value && t == 0) rb_mma_c(acc[p], av, B[ky + p], cinit);
                    else rb_mma(acc[p], av, B[ky + p]);
                });
                // 1 MFMA, the B line, 1 MFMA, the A fragment, the rest
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            } else {
                uint4 Bn[8];
                const int o = boff(std::integral_constant<int, (g + 1 < 12 ? g + 1 : 0)>{});
                rb_for<0, 8>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    rb_mma(acc[p], av, B[2 + p]);
                    Bn[p] = *(const uint4*)line(P, o);
                });
                rb_for<0, 8>([&](auto P) { B[decltype(P)::value] = Bn[decltype(P)::value]; });
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    const char* const tin = smem + rh * 8 * RB_IN_LS;
    auto line_in = [&](auto L, int o) { return tin + decltype(L)::value * RB_IN_LS + o; };
    const char* const tm = smem + RB_M_OFF + (8 * rh - 2) * RB_ML;      // window line L >= 2 of this wave's 10 (rh == 0: lines 0, 1 are carried)
    const int col_x = lx - 1;                                          // + x0 = image column of this lane's intermediate pixel

    int it = it0;
    bool pro = true;
    int cp = 0;
    [[maybe_unused]] int trk = 0;
    for (;;) {
        int img, x0, row0;
        pos_of(it, pro, img, x0, row0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous step's staged outputs (and the biases) are in LDS
        RB_STAMP(wave, trk, 0);
        asm volatile("s_barrier" ::: "memory");                 // A: the input window has landed
        RB_STAMP(wave, trk, 1);
        {
            f16x_t c1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4_t q = *(const f4_t*)(smem + RB_BIAS_OFF + (cs * 32 + g * 8 + hi * 4) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) c1[g * 4 + j] = q[j];
            }
            if (!pro) conv_phase(line_in, w1, w2, std::true_type{}, c1);
            else {
                // the other seven accumulators are dead here; say so (an empty asm "defines" them), or the allocator carries their old
                // values through this path in scratch
                asm volatile("" : "=v"(acc[1]), "=v"(acc[2]), "=v"(acc[3]), "=v"(acc[4]), "=v"(acc[5]), "=v"(acc[6]), "=v"(acc[7]));
                // chain-opening step: only the two carried lines (block rows 14, 15) are wanted -- wave (cs, rh) computes row 14 + rh
                // with ONE accumulator: 36 x (A fragment + MFMA) instead of 36 x 8 MFMAs (18 700 -> ~7 000 cycles per chain)
                const char* const tp = smem + (14 + rh) * RB_IN_LS;
                uint4 Bc[3], Bn[3];
                rb_for<0, 3>([&](auto L) { Bc[decltype(L)::value] = *(const uint4*)(tp + decltype(L)::value * RB_IN_LS + boff(std::integral_constant<int, 0>{})); });
                rb_for<0, 12>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    if constexpr (g + 1 < 12) {
                        const int o = boff(std::integral_constant<int, g + 1>{});
                        rb_for<0, 3>([&](auto L) { Bn[decltype(L)::value] = *(const uint4*)(tp + decltype(L)::value * RB_IN_LS + o); });
                    }
                    rb_for<0, 3>([&](auto KY) {
                        constexpr int t = 3 * g + decltype(KY)::value;
                        const uint4 av = A[t % RB_DEPTH];
                        A[t % RB_DEPTH] = a_load(w1, std::integral_constant<int, (t + RB_DEPTH) % RB_NSTEP>{});
                        if constexpr (t == 0) rb_mma_c(acc[0], av, Bc[0], c1);
                        else rb_mma(acc[0], av, Bc[decltype(KY)::value]);
                    });
                    rb_for<0, 3>([&](auto L) { Bc[decltype(L)::value] = Bn[decltype(L)::value]; });
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
#if defined(DEMFI_TRACE) && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
#endif
        RB_STAMP(wave, trk, 2);
        asm volatile("s_barrier" ::: "memory");                 // B: the helpers hold the previous outputs in registers
        RB_STAMP(wave, trk, 3);
        {
            // conv1 epilogue: ReLU, fp16 -> M lines (rows 14, 15 also to the carry slot the NEXT step reads); the accumulators then
            // restart at bias2 + identity (the row's identity is read from the input window before its conversions are issued).  A
            // lone wave issues one VALU instruction per ~10 cycles (tools/microbench/valu_rate.hip), so the epilogue is its instruction
            // count: conv2's zero padding -- M is zero outside the image -- is therefore not a mask on every value but a second pass
            // that zeroes the few lines / columns concerned, on the ~9 % of the steps that touch the image border.
            const bool edge = row0 + 1 < 0 || row0 + RB_R + 1 > H || x0 - 1 < 0 || x0 + 31 > W;
            if (pro) {                                           // the opening step's single row -> carry line rh of the slot the next step reads
                const bool ok = (unsigned)(row0 + 15 + rh) < (unsigned)H && (unsigned)(x0 + col_x) < (unsigned)W;
                char* const cr = smem + RB_C_OFF + (cp ^ 1) * 2 * RB_ML + rh * RB_ML;
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    h8_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = (half_t)acc[0][(2 * m2) * 4 + j];
                        o[4 + j] = (half_t)acc[0][(2 * m2 + 1) * 4 + j];
                    }
                    const h8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
                    o = __builtin_elementwise_max(o, z);
                    u4_t ob = __builtin_bit_cast(u4_t, o);
                    ob &= ok ? 0xffffffffu : 0u;
                    *(u4_t*)(cr + soff[m2]) = ob;
                }
            }
            f16x_t c2;
            if (!pro) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f4_t q = *(const f4_t*)(smem + RB_BIAS_OFF + (64 + cs * 32 + g * 8 + hi * 4) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) c2[g * 4 + j] = q[j];
                }
            }
            char* const mrow = smem + RB_M_OFF + rh * 8 * RB_ML;
            char* const crow = smem + RB_C_OFF + (cp ^ 1) * 2 * RB_ML;
            rb_for<0, 8>([&](auto P) {
                constexpr int p = decltype(P)::value;
                if (pro) return;                                 // per row (see the note on basic blocks above the kernel's epilogue)
                u4_t idr[2];
                idr[0] = *(const u4_t*)(tin + p * RB_IN_LS + ioff[0]);
                idr[1] = *(const u4_t*)(tin + p * RB_IN_LS + ioff[1]);
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    h8_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = (half_t)acc[p][(2 * m2) * 4 + j];
                        o[4 + j] = (half_t)acc[p][(2 * m2 + 1) * 4 + j];
                    }
                    const h8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
                    o = __builtin_elementwise_max(o, z);
                    const u4_t ob = __builtin_bit_cast(u4_t, o);
                    *(u4_t*)(mrow + p * RB_ML + soff[m2]) = ob;
                    if (p >= 6 && rh == 1) *(u4_t*)(crow + (p - 6) * RB_ML + soff[m2]) = ob;
                }
                {
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) {
                        const u4_t r = idr[m2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            acc[p][(2 * m2) * 4 + 2 * q] = rb_mix_lo(r[q], c2[(2 * m2) * 4 + 2 * q]);
                            acc[p][(2 * m2) * 4 + 2 * q + 1] = rb_mix_hi(r[q], c2[(2 * m2) * 4 + 2 * q + 1]);
                            acc[p][(2 * m2 + 1) * 4 + 2 * q] = rb_mix_lo(r[2 + q], c2[(2 * m2 + 1) * 4 + 2 * q]);
                            acc[p][(2 * m2 + 1) * 4 + 2 * q + 1] = rb_mix_hi(r[2 + q], c2[(2 * m2 + 1) * 4 + 2 * q + 1]);
                        }
                    }
                }
            });
            if (edge && !pro) {
                // LDS operations of one wave complete in order: these zeros land on top of the values this lane has just written
                const bool col_out = !((unsigned)(x0 + col_x) < (unsigned)W);
                const u4_t zz = {0u, 0u, 0u, 0u};
                rb_for<0, 8>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    const bool row_out = !((unsigned)(row0 + 1 + rh * 8 + p) < (unsigned)H);
                    if (row_out || col_out) {
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) {
                            *(u4_t*)(mrow + p * RB_ML + soff[m2]) = zz;
                            if (p >= 6 && rh == 1) *(u4_t*)(crow + (p - 6) * RB_ML + soff[m2]) = zz;
                        }
                    }
                });
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RB_STAMP(wave, trk, 4);
        asm volatile("s_barrier" ::: "memory");                 // C: M is complete, the input window is free
        RB_STAMP(wave, trk, 5);
        if (!pro) {
            const char* const t01 = rh == 0 ? smem + RB_C_OFF + cp * 2 * RB_ML : smem + RB_M_OFF + 6 * RB_ML;
            auto line_m = [&](auto L, int o) {
                constexpr int l = decltype(L)::value;
                if constexpr (l < 2) return t01 + l * RB_ML + o;
                else return tm + l * RB_ML + o;
            };
            const f16x_t none = {};
            conv_phase(line_m, w2, w1, std::false_type{}, none);
        }
#if defined(DEMFI_TRACE) && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
#endif
        RB_STAMP(wave, trk, 6);
        asm volatile("s_barrier" ::: "memory");                 // D: nobody reads M any more -> its 16 lines take the staged outputs
        RB_STAMP(wave, trk, 7);
        if (!pro) {
            char* const srow = smem + RB_M_OFF + rh * 8 * RB_ML;
            rb_for<0, 8>([&](auto P) {
                constexpr int p = decltype(P)::value;
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    h8_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = (half_t)acc[p][(2 * m2) * 4 + j];
                        o[4 + j] = (half_t)acc[p][(2 * m2 + 1) * 4 + j];
                    }
                    *(u4_t*)(srow + p * RB_ML + soff[m2]) = __builtin_bit_cast(u4_t, o);
                }
            });
        }
        RB_STAMP(wave, trk, 8);
        ++trk;
        cp ^= 1;
        if (pro) pro = false;
        else {
            ++it;
            if (it >= it1) break;
            pro = opens_chain(it);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                     // F
}

// the pair (h1, h2) is one residual block this kernel can run: two 3x3 64 -> 64 fp16 convolutions in the packing of the 64-channel
// persistent kernels, ReLU between them, conv2's residual = conv1's input, nothing else in the epilogues
bool rb_layer_ok(const demfi_conv* h)
{
    if (h->dtype != DEMFI_F16 || h->stride != 1 || h->kh != 3 || h->kw != 3 || h->pad_y != 1 || h->pad_x != 1) return false;
    if (h->n_chunks != 1 || h->n_pieces != 1 || h->chunks[0].nks != 4 || h->rec_bytes != 128 || h->nco != 2 || h->cout_pad != 64) return false;
    if (!h->cout_perm || !h->zero_page || h->inH != h->H || h->inW != h->W || !h->wpack || !h->bias) return false;
    const demfi_piece& p = h->pieces[0];
    if (!p.fat || p.nch != 64 || p.up_shift != 0 || !p.v.ptr || p.v.is_f32 || p.v.sc != 1) return false;
    const int sg = h->sub_seg[0];
    if (sg < 0 || h->sub_seg[1] != sg || h->oct_ch[4] != h->oct_ch[0] + 32) return false;
    // exactly what the fused kernel reproduces and nothing it would silently drop: ONE segment that owns all eight octets
    // as one run of 64 channels, no packed copy, no uint8 sink, no aux view
    if (h->n_segs != 1 || h->pack.ptr || h->u8_sink) return false;
    for (int o = 0; o < 8; ++o)
        if (h->oct_seg[o] != sg || h->oct_n[o] != 8 || h->oct_ch[o] != h->oct_ch[0] + 8 * o) return false;
    const demfi_seg& seg = h->segs[sg];
    if (seg.aux.ptr) return false;
    if (seg.mode != DEMFI_MODE_STORE || seg.scale != 1 || seg.dy || seg.dx || !seg.dst.ptr || seg.dst.is_f32 || seg.dst.sc != 1) return false;
    return true;
}

}  // namespace

extern "C" int demfi_resblock_eligible(const demfi_conv* h1, const demfi_conv* h2)
{
    if (!h1 || !h2 || !rb_layer_ok(h1) || !rb_layer_ok(h2)) return 0;
    if (h1->H != h2->H || h1->W != h2->W || h1->batch != h2->batch) return 0;
    const demfi_seg& s1 = h1->segs[h1->sub_seg[0]];
    const demfi_seg& s2 = h2->segs[h2->sub_seg[0]];
    if (s1.act != DEMFI_ACT_RELU || s1.res.ptr) return 0;
    if (s2.act != DEMFI_ACT_NONE || !s2.res.ptr || s2.res.is_f32) return 0;
    const demfi_view& x = h1->pieces[0].v;
    // conv2 reads what conv1 wrote, its residual is conv1's input
    const demfi_view& t = h2->pieces[0].v;
    if ((const char*)t.ptr != (const char*)s1.dst.ptr + (int64_t)h1->oct_ch[0] * 2 || t.sx != s1.dst.sx || t.sy != s1.dst.sy || t.sb != s1.dst.sb) return 0;
    if ((const char*)s2.res.ptr + (int64_t)h2->oct_ch[0] * 2 != (const char*)x.ptr || s2.res.sx != x.sx || s2.res.sy != x.sy || s2.res.sb != x.sb || s2.res.sc != 1) return 0;
    if (s2.dst.ptr == x.ptr) return 0;                           // in-place is not possible: neighbouring strips read the halo
    // 32-bit per-lane offsets inside a window / an output block
    if (x.sy * 2 * 20 + x.sx * 2 * 40 >= (int64_t)1 << 31 || s2.dst.sy * 2 * 20 + s2.dst.sx * 2 * 40 >= (int64_t)1 << 31) return 0;
    return 1;
}

extern "C" int demfi_resblock3x3_c64(const demfi_conv* h1, const demfi_conv* h2, void* stream)
{
    if (!demfi_resblock_eligible(h1, h2))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_resblock3x3_c64: not a fusable residual block (two 3x3 64->64 fp16 layers in persistent-kernel "
                                              "packing, ReLU between, conv2.res == conv1 input)");
    RbArgs a;
    const demfi_view& x = h1->pieces[0].v;
    const demfi_seg& s2 = h2->segs[h2->sub_seg[0]];
    a.src = (const char*)x.ptr; a.s_sx = x.sx * 2; a.s_sy = x.sy * 2; a.s_sb = x.sb * 2;
    a.dst = (char*)s2.dst.ptr + (int64_t)h2->oct_ch[0] * 2; a.d_sx = s2.dst.sx * 2; a.d_sy = s2.dst.sy * 2; a.d_sb = s2.dst.sb * 2;
    a.w1 = (const char*)h1->wpack; a.w2 = (const char*)h2->wpack;
    a.b1 = h1->bias; a.b2 = h2->bias;
    a.zeros = (const char*)h1->zero_page;
    a.H = h1->H; a.W = h1->W; a.batch = h1->batch;
    a.n_strips = (a.W + RB_OW - 1) / RB_OW;
    a.n_rsteps = (a.H + RB_R - 1) / RB_R;
    const int64_t total = (int64_t)a.n_strips * a.n_rsteps * a.batch;
    if (total <= 0 || total >= (int64_t)1 << 30) return demfi_set_error(DEMFI_ERR_ARG, "demfi_resblock3x3_c64: empty or oversized launch");
    DEMFI_LDS_ATTR(resblock3x3_c64_kernel);
    const int grid = total >= 256 ? 256 : (int)total;
    hipLaunchKernelGGL(resblock3x3_c64_kernel, dim3(grid), dim3(RB_NTHREADS), RB_LDS, (hipStream_t)stream, a);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

#ifdef DEMFI_TRACE
// copies the trace to host memory and clears it (trace build only)
extern "C" int demfi_rb_trace_dump(unsigned long long* out, int64_t n)
{
    const int64_t have = (int64_t)RB_TR_WGS * RB_TR_WAVES * RB_TR_STEPS * RB_TR_STAMPS;
    if (n != have) return demfi_set_error(DEMFI_ERR_ARG, "demfi_rb_trace_dump: expected %lld entries", (long long)have);
    DEMFI_HIP_CHECK(hipDeviceSynchronize());
    DEMFI_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rb_trace), have * 8));
    static const std::vector<unsigned long long> zeros(have, 0ull);
    DEMFI_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_rb_trace), zeros.data(), have * 8));
    return DEMFI_OK;
}
#endif
