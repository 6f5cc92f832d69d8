// SepConvGRU half-step on the matrix cores with the update gate kept on chip (round 6).
//
// One half-step of SepConvGRU (DeMFInet.py:838-857; horizontal 1x5: 844-849, vertical 5x1: 851-856) is
//     z = sigmoid(convz([h, x]))      r = sigmoid(convr([h, x]))      q = tanh(convq([r * h, x]))      h' = (1 - z) h + z q
// Rounds 1-5 ran it as two launches of conv_sep5_c128_persist_kernel (conv.hip): z | r (writes z and r * h), then q (reads z, r * h,
// h, x).  z makes a round trip through HBM (256 of the pair's 1 152 B per pixel) and the q launch sits on the memory wall
// (640 B/px at 4.9 TB/s, profiles/r03_gru_hbm_bytes_pmc.txt).  This file splits the step the other way round:
//     launch R  : r * h                         (reads h, x; writes r * h)                           384 B/px
//     launch ZQ : z AND q AND the blend         (reads h, r * h, x; writes h')                       512 B/px
// z never leaves the CU: the z waves hand it to the q waves through LDS (rounded to fp16 exactly as the stored z was).
//
// Tile orientation: the MFMA's 32-pixel axis runs ACROSS the filter axis, lines run ALONG it, so a filter
// tap shifts the LINE index: the B fragment of input line l serves output line p at tap l - p, one ds_read_b128 feeds up to five
// MFMAs (the 3x3 kernels' ky trick; the old kernel shifted the pixel axis: one fragment per tap).  A wave owns 8 lines x 32 couts
// = eight 32x32 accumulators; per 16-channel k-step it reads 12 line fragments and 5 A fragments for 40 MFMAs.  The A (weight)
// fragments do not live in LDS: one global_load_dwordx4 per (tap, k-step) from the packed, L2-resident weights into a register ring
// (resblock.hip's scheme), which is what frees the LDS for whole 64-channel windows.
//
// ZQ: a workgroup = 4 MFMA waves (z cout halves, q cout halves) + 4 helper waves, tile = 32 pixels x 8 lines, three window pieces
// of 12 lines x 32 px x 128 B in LDS (h | r*h | x = 144 KiB), each with a fixed address and refilled for the next tile as soon as the
// last wave is done with it.  Phase A: the z waves contract h, the q waves r*h; phase B: both contract x.  Then z -> sigmoid ->
// fp16 -> LDS (into the x piece, dead by then), the q waves tanh their accumulators, blend with z and h (h prefetched from global)
// and stage h' as fp16 in the same place; the helper waves store it as whole 128-byte lines (streaming) and issue the DMA.
// R: tile = 32 pixels x 16 lines, wave = cout half x line half, pieces x | h of 20 lines (160 KiB), sigmoid * h from the h window.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

constexpr int G_LS = 32 * 128;                                   // bytes per window line (32 records of 64 channels)
constexpr int G_NH = 4;                                          // helper waves
constexpr int G_NTHREADS = 256 + 64 * G_NH;
#ifndef DEMFI_ZQS_PRIME_EARLY
#define DEMFI_ZQS_PRIME_EARLY 0
#endif
#ifndef DEMFI_GRU_DEPTH
#define DEMFI_GRU_DEPTH 4                                        // A prefetch distance in steps of 8 MFMAs
#endif
constexpr int G_DEPTH = DEMFI_GRU_DEPTH;
constexpr int G_NSTEP = 20;                                      // (k-step, tap) steps per 64-channel piece
static_assert(G_NSTEP % G_DEPTH == 0, "static ring indices");
enum { GM_ZQ = 0, GM_R = 1, GM_ZQS = 2 };                        // ZQS: ZQ with both gates of a cout block in one wave (below)
template <int MODE> struct GCfg;
template <> struct GCfg<GM_ZQ>  { static constexpr int TL = 8, WL = 12, NPIECE = 3, LDS = 3 * 12 * G_LS; };
template <> struct GCfg<GM_R>   { static constexpr int TL = 16, WL = 20, NPIECE = 2, LDS = 2 * 20 * G_LS; };
template <> struct GCfg<GM_ZQS> { static constexpr int TL = 8, WL = 12, NPIECE = 3, LDS = 3 * 12 * G_LS + 512; };   // + 128 bias floats
static_assert(GCfg<GM_R>::LDS <= 160 * 1024 && GCfg<GM_ZQS>::LDS <= 160 * 1024, "LDS budget");
constexpr int ZS_D = 8;                                          // ZQS: A ring depth in fragments
constexpr int ZS_NA = 80;                                        // ZQS: A fragments per tile and wave: 20 (z . h) + 20 (q . r*h) + 40 (z, q alternating . x)
static_assert(ZS_NA % ZS_D == 0, "static ring indices");

#ifdef DEMFI_TRACE
constexpr int GT_WGS = 32, GT_WAVES = 8, GT_TILES = 24, GT_STAMPS = 10;
__device__ unsigned long long g_gru_trace[GT_WGS * GT_WAVES * GT_TILES * GT_STAMPS];
#define G_STAMP(wave_, k_, i_)                                                                                        \
    do {                                                                                                              \
        if (blockIdx.x < GT_WGS && (k_) < GT_TILES && (threadIdx.x & 63) == 0)                                        \
            g_gru_trace[((blockIdx.x * GT_WAVES + (wave_)) * GT_TILES + (k_)) * GT_STAMPS + (i_)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define G_STAMP(wave_, k_, i_) do { } while (0)
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void g_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        g_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ void g_mma(f16x_t& acc, const uint4& a, const uint4& b)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void g_mma_c(f16x_t& acc, const uint4& a, const uint4& b, const f16x_t& c)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
}
// c - (fp16 half of a packed pair) in one VALU op: tanh(.) - h
__device__ __forceinline__ float g_sub_lo(float c, unsigned a)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}
__device__ __forceinline__ float g_sub_hi(float c, unsigned a)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}

// c - b * (fp16 half of a packed pair): (b - 1) - (b + 1) h
__device__ __forceinline__ float g_nfma_lo(float b, unsigned h, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, -%1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(b), "v"(c));
#endif
    return d;
}
__device__ __forceinline__ float g_nfma_hi(float b, unsigned h, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, -%1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(b), "v"(c));
#endif
    return d;
}

// q - h on two packed fp16 operands (low / high halves) in one VALU op, fp32 result
__device__ __forceinline__ float g_sub16_lo(unsigned q, unsigned h)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(q));
#endif
    return d;
}
__device__ __forceinline__ float g_sub16_hi(unsigned q, unsigned h)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(q));
#endif
    return d;
}
// fp16(a * b + c) with c = low / high half of a packed pair, written to the low / high half of pk: two of them make one packed dword
__device__ __forceinline__ void g_fma16_lo(unsigned& pk, float a, float b, unsigned c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "+v"(pk) : "v"(a), "v"(b), "v"(c));
#endif
}
__device__ __forceinline__ void g_fma16_hi(unsigned& pk, float a, float b, unsigned c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(pk) : "v"(a), "v"(b), "v"(c));
#endif
}

struct GPiece { const char* ptr; int64_t sl, sp, sb; };          // source tensor: byte strides along the filter axis, across it, per image
struct GArgs {
    GPiece h, rh, x;                                             // ZQ: h | r*h | x;  R: h | (unused) | x
    char* dst; int64_t d_sl, d_sp, d_sb;                         // h' (ZQ) or r*h (R), at its first channel
    const char* w0[2];                                           // packed A fragments of layer 0 (z resp. r): chunk 0 (h), chunk 1 (x)
    const char* w1[2];                                           // ZQ: layer 1 (q): chunk 0 (r*h), chunk 1 (x)
    const float* b0; const float* b1;                            // packed cout order
    const char* zeros;
    int Llen, Plen, batch, n_pt, n_ls;                           // extent along / across the filter axis; tiles across, steps along
};

template <int MODE>
__global__ __launch_bounds__(G_NTHREADS, 1) void gru_sep5_kernel(const GArgs a)
{
    using Cfg = GCfg<MODE>;
    constexpr int TL = Cfg::TL, WL = Cfg::WL, PIECE = WL * G_LS;
    constexpr int NI = WL * 4, NIW = NI / G_NH;                   // DMA instructions per piece (8 records each) / per helper: one per line
    static_assert(NIW == WL, "helper dw owns pixels 8 dw .. 8 dw + 7 of every line");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_img = a.n_pt * a.n_ls;
    const int total = per_img * a.batch;
    int it0, it1;
    {
        // contiguous run of items (image, tile across, step along -- along fastest) per workgroup; the workgroups of an XCD
        // (blockIdx % 8) share a contiguous band, so the 4 halo lines two consecutive steps share meet in one L2
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
    auto pos_of = [&](int it, int& img, int& P0, int& L0) {
        img = it / per_img;
        const int rem = it - img * per_img;
        const int pt = rem / a.n_ls;
        P0 = pt * 32;
        L0 = (rem - pt * a.n_ls) * TL;
    };
    // LDS pieces (fixed addresses).  ZQ: 0 = h, 1 = r*h, 2 = x (later: z, then the staged h');  R: 0 = x, 1 = h
    char* const S0 = smem;
    char* const S1 = smem + PIECE;
    char* const S2 = smem + 2 * PIECE;                           // ZQ only

    if constexpr (MODE == GM_ZQS) {
        if (tid < 128) ((float*)(smem + 3 * PIECE))[tid] = tid < 64 ? a.b0[tid] : a.b1[tid - 64];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // written before this wave's first (raw) barrier
    }
    if (wave >= 4) {
        // ================= helper waves: window DMA (+ ZQ: the global stores of the staged outputs) ======================
        const int dw = wave - 4;
        __builtin_assume(dw >= 0 && dw < G_NH);
        const int px = dw * 8 + (lane >> 3);                     // pixel of this lane in every DMA instruction / staged line
        const unsigned slot16 = (unsigned)(((lane & 7) ^ ((px >> 1) & 7)) << 4);
        // instruction k of helper dw = window line k, records 8 dw .. 8 dw + 7 -> LDS bytes [(4 k + dw) 1024, + 1024)
        // carry: the tile continues the previous one along the filter axis (same image, same pixels, TL lines further): its first 4
        // window lines ARE the previous window's last 4 -- copied inside LDS (this helper's chunk of each line: 4 ds_read + 4 ds_write)
        // instead of fetched again.  A helper's VMEM instruction is the scarce thing (130-400 cycles each beside the MFMA waves) and the
        // re-fetched halo lines missed the XCD's L2 (32 windows of 144 KiB per XCD): -1/3 of the DMA instructions, -1/3 of the fabric reads.
        auto issue_piece = [&](const GPiece& pc, char* lds, int img, int P0, int L0, bool carry) {
            const char* base = pc.ptr + (int64_t)img * pc.sb + (int64_t)(L0 - 2) * pc.sl + (int64_t)P0 * pc.sp;      // wave-uniform
            const unsigned voff = (unsigned)(px * pc.sp) + slot16;
            const bool pok = P0 + px < a.Plen;
            u4_t halo[4];
            if (carry) {
#pragma unroll
                for (int j = 0; j < 4; ++j) halo[j] = *(const u4_t*)(lds + (4 * (TL + j) + dw) * 1024 + lane * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // in registers before the DMA below may overwrite lines TL .. TL + 3
            }
            if (L0 - 2 >= 0 && L0 - 2 + WL <= a.Llen && P0 + 32 <= a.Plen) {
                // uniform line base (SALU) + ONE 32-bit lane offset: the saddr form, no VALU per instruction -- the helpers share their
                // SIMDs with the MFMA waves and a starved helper needed ~600 cycles per instruction with 64-bit per-lane addresses
#pragma unroll
                for (int k = 0; k < NIW; ++k) {
                    if (k < 4 && carry) continue;                // wave-uniform
                    const char* lb = base + (int64_t)k * pc.sl;
                    lb = (const char*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)((uint64_t)lb >> 32)) << 32) |
                                       (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(uint64_t)lb));   // uniform already: folds away, pins lb to SGPRs
                    unsigned v = voff;
                    asm volatile("" : "+v"(v));
                    __builtin_amdgcn_global_load_lds((const DEMFI_GLOBAL void*)(lb + v),
                                                     (__attribute__((address_space(3))) void*)(lds + (4 * k + dw) * 1024), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NIW; ++k) {
                    if (k < 4 && carry) continue;
                    const int l = L0 - 2 + k;
                    const char* g = (pok && l >= 0 && l < a.Llen) ? base + (int64_t)k * pc.sl + voff : a.zeros;
                    __builtin_amdgcn_global_load_lds((const DEMFI_GLOBAL void*)g,
                                                     (__attribute__((address_space(3))) void*)(lds + (4 * k + dw) * 1024), 16, 0, 0);
                }
            }
            if (carry) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *(u4_t*)(lds + (4 * j + dw) * 1024 + lane * 16) = halo[j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // written before this wave arrives at the barrier that hands the piece over
            }
        };
        int it = it0;
        int img, P0, L0;
        pos_of(it, img, P0, L0);
        [[maybe_unused]] int trk = 0;
        if constexpr (MODE != GM_R) {
            // staged outputs (in piece 2): line r = bytes [4096 r, +4096), helper dw owns its chunk dw = pixels 8 dw .. 8 dw + 7
            const unsigned dlane = (unsigned)(px * a.d_sp) + slot16;
            issue_piece(a.h, S0, img, P0, L0, false);
            issue_piece(a.rh, S1, img, P0, L0, false);
            issue_piece(a.x, S2, img, P0, L0, false);
            // A helper's VMEM instruction takes ~400 cycles while the MFMA waves of its SIMD run a matrix phase and 130-220 while they are in
            // their epilogues, and it slows the matrix phase it runs beside (profiles/r06_gru_phase_trace.txt, r06_resblock_store_timing_ab.txt).
            // Per tile: x (8 DMA instructions; 12 at a strip start) under phase A -- it needs the staged outputs read first --, the next tile's h
            // (8) under phase B, its r*h (8) under the sigmoid / tanh pass, the previous tile's 8 output stores under the blend (round 6: they
            // ran under phase A, which took 6 950 cycles with them and 5 870 without).
            // Barrier E doubles as "the next tile's h, r*h have landed": no barrier in front of phase A.
            u4_t stage[TL];
            bool have_prev = false;
            int pimg = 0, pP0 = 0, pL0 = 0;
            auto do_stores = [&](int simg, int sP0, int sL0) {
                char* const obase = a.dst + (int64_t)simg * a.d_sb + (int64_t)sL0 * a.d_sl + (int64_t)sP0 * a.d_sp;     // wave-uniform
                const bool pok = sP0 + px < a.Plen;
                if (sL0 + TL <= a.Llen) {
                    if (pok) {
#pragma unroll
                        for (int r = 0; r < TL; ++r) __builtin_nontemporal_store(stage[r], gp<u4_t>(obase + (int64_t)r * a.d_sl + dlane));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < TL; ++r)
                        if (pok && sL0 + r < a.Llen) __builtin_nontemporal_store(stage[r], gp<u4_t>(obase + (int64_t)r * a.d_sl + dlane));
                }
                return sL0 + TL <= a.Llen && sP0 + 32 <= a.Plen;  // exactly TL store instructions were issued (vmcnt bookkeeping)
            };
            // loads and stores count together, in order
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");      // the first tile's h, r*h have landed (x may be in flight)
            G_STAMP(wave, trk, 0);
            asm volatile("s_barrier" ::: "memory");             // A0
            for (;;) {
                // the MFMA waves are in phase A of this tile; this tile's x was issued behind barrier E of the previous one
                const bool more = it + 1 < it1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // x has landed (older: the stores of tile k - 2)
                G_STAMP(wave, trk, 1);
                asm volatile("s_barrier" ::: "memory");         // B: x has landed; every MFMA wave is done with h, r*h
                int nimg = img, nP0 = P0, nL0 = L0;
                bool carry = false;
                if (more) {                                      // the next tile's h (8 DMA instructions, 12 at a strip start) under phase B ...
                    pos_of(it + 1, nimg, nP0, nL0);
                    carry = nimg == img && nP0 == P0 && nL0 == L0 + TL;
                    issue_piece(a.h, S0, nimg, nP0, nL0, carry);
                }
                G_STAMP(wave, trk, 2);
                asm volatile("s_barrier" ::: "memory");         // C: every MFMA wave is done with x
                // ... r*h (8) under the sigmoid / tanh pass.  Both there (16) delay barrier D by ~1 400 cycles: measured, profiles/r06_notes.md
                if (more) issue_piece(a.rh, S1, nimg, nP0, nL0, carry);
                G_STAMP(wave, trk, 3);
                if constexpr (MODE == GM_ZQ) asm volatile("s_barrier" ::: "memory");         // D: q~ is in LDS  (ZQS: wave-local epilogue, no D)
                bool exact = false;
                if (have_prev) exact = do_stores(pimg, pP0, pL0);                          // the previous tile's outputs (8 stores) under the blend
                if (exact) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TL) : "memory");      // the next tile's h, r*h (older than the stores) have landed:
                else       asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // E tells the MFMA waves
                G_STAMP(wave, trk, 4);
                asm volatile("s_barrier" ::: "memory");         // E: h' is staged
                {
                    const char* sbp = S2 + dw * 1024 + lane * 16;
#pragma unroll
                    for (int r = 0; r < TL; ++r) stage[r] = *(const u4_t*)(sbp + r * G_LS);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                // this helper's DMA instructions overwrite exactly the chunks it has just read
                if (more) issue_piece(a.x, S2, nimg, nP0, nL0, carry);   // lines 8 .. 11 of the x piece survived the staging (lines 0 .. 7)
                G_STAMP(wave, trk, 5);
                ++trk;
                have_prev = true; pimg = img; pP0 = P0; pL0 = L0;
                if (!more) break;
                ++it; img = nimg; P0 = nP0; L0 = nL0;
            }
            do_stores(pimg, pP0, pL0);
        } else {
            issue_piece(a.x, S0, img, P0, L0, false);
            issue_piece(a.h, S1, img, P0, L0, false);
            bool h_carried = false;                              // the h piece in flight was issued with the halo carried (4 instructions fewer)
            for (;;) {
                const bool more = it + 1 < it1;
                // outstanding, in order: x | h  ->  x has landed when only h's instructions are left
                if (h_carried) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW - 4) : "memory");
                else           asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
                G_STAMP(wave, trk, 0);
                asm volatile("s_barrier" ::: "memory");         // A: x has landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                G_STAMP(wave, trk, 1);
                asm volatile("s_barrier" ::: "memory");         // B: h has landed; every MFMA wave is done with x
                int nimg = img, nP0 = P0, nL0 = L0;
                bool carry = false;
                if (more) {
                    pos_of(it + 1, nimg, nP0, nL0);
                    carry = nimg == img && nP0 == P0 && nL0 == L0 + TL;
                    issue_piece(a.x, S0, nimg, nP0, nL0, carry);
                }
                G_STAMP(wave, trk, 2);
                asm volatile("s_barrier" ::: "memory");         // C: epilogues done, h is free
                if (more) issue_piece(a.h, S1, nimg, nP0, nL0, carry);
                h_carried = carry;
                G_STAMP(wave, trk, 3);
                ++trk;
                if (!more) break;
                ++it; img = nimg; P0 = nP0; L0 = nL0;
            }
        }
        return;
    }

    // ================= MFMA waves, ZQS ===========================================================================
    // Both gates of a cout block in one wave: wave = cout half cs x line half lh, four z and four q accumulators (4 lines x 32 couts each).
    // What that buys over ZQ (where two waves own z and two own q, 8 lines each): the epilogue is wave-local -- no q~ exchange through LDS and
    // no barrier D; z and q~ of an output meet in one lane, so h' = h + ((b - 1) - h (b + 1)) / ((b + 1)(1 + a)) with a = e^-z', b = e^2q' takes
    // THREE transcendentals per output instead of four (the pass sits on the quarter-rate transcendental unit); and the blend runs on all four
    // SIMDs instead of the z waves' two.  What it costs: one A fragment per four MFMAs instead of eight in the h and r*h phases (the x phase feeds
    // both gates from the same B fragments: LDS reads per tile unchanged) -- measured beforehand with dummy loads in a second ring: + 4 %.
    if constexpr (MODE == GM_ZQS) {
        const int hi = lane >> 5, lx = lane & 31;
        const int cs = wave & 1, lh = wave >> 1;
        const unsigned lane16 = lane * 16;
        const char* const wz0 = a.w0[0] + cs * 1024;             // z . h
        const char* const wq0 = a.w1[0] + cs * 1024;             // q . r*h
        const char* const wz1 = a.w0[1] + cs * 1024;             // z . x
        const char* const wq1 = a.w1[1] + cs * 1024;             // q . x
        const int boff0 = lx * 128 + ((hi ^ ((lx >> 1) & 7)) << 4);
        auto boff = [&](auto KS) {
            int b = boff0;
            asm volatile("" : "+v"(b));
            return b ^ (decltype(KS)::value << 5);
        };
        int soff[2];
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) soff[m2] = lx * 128 + (((cs * 4 + m2 * 2 + hi) ^ ((lx >> 1) & 7)) << 4);
        auto a_loadn = [&](auto N_) {                            // fragment nn of the tile's A stream
            constexpr int nn = decltype(N_)::value;
            constexpr int sel = nn < 20 ? 0 : nn < 40 ? 1 : 2 + ((nn - 40) & 1);
            constexpr int t = nn < 20 ? nn : nn < 40 ? nn - 20 : (nn - 40) >> 1;
            constexpr int ks = t / 5, tap = t % 5;
            const char* wb = sel == 0 ? wz0 : sel == 1 ? wq0 : sel == 2 ? wz1 : wq1;
            asm volatile("" : "+s"(wb));
            unsigned l16 = lane16;
            asm volatile("" : "+v"(l16));
            return __builtin_bit_cast(uint4, *gcp<u4_t>(wb + (unsigned)(((tap * 4 + ks) * 2) * 1024 + l16)));
        };
        uint4 A[ZS_D];
        uint4 B[8];
        f16x_t accz[4], accq[4];
        auto prime = [&]() { g_for<0, ZS_D>([&](auto T) { A[decltype(T)::value] = a_loadn(T); }); };
        auto b_init = [&](const char* tb) {
            const int o = boff(std::integral_constant<int, 0>{});
            g_for<0, 4>([&](auto R) { B[decltype(R)::value] = *(const uint4*)(tb + decltype(R)::value * G_LS + o); });
        };
        // one 64-channel window: 4 k-steps x 5 taps; WHICH = 0: z (4 MFMAs per step), 1: q, 2: both (8 MFMAs, two A fragments).  This wave's
        // B window = lines tb + 0 .. 7: step (ks, tap) multiplies output line p with line tap + p; line 4 + tap arrives during tap; during tap 4
        // the next k-step's lines 0 .. 3 (B[0 .. 3] are dead by then) -- or, at the end of the h window, the first lines of the r*h window
        auto phase = [&](const char* tb, const char* tbn, auto NB_, auto WHICH_, auto HASNEXT_) {
            constexpr int NB = decltype(NB_)::value, WHICH = decltype(WHICH_)::value;
            constexpr bool DUAL = WHICH == 2, HASNEXT = decltype(HASNEXT_)::value;
            constexpr int PER = DUAL ? 2 : 1;
            g_for<0, G_NSTEP>([&](auto T_) {
                constexpr int t = decltype(T_)::value;
                constexpr int ks = t / 5, tap = t % 5;
                constexpr int n0 = NB + t * PER, n1 = n0 + PER - 1;
                const uint4 av0 = A[n0 % ZS_D];
                const uint4 av1 = A[n1 % ZS_D];
                if constexpr (tap < 4) B[4 + tap] = *(const uint4*)(tb + (4 + tap) * G_LS + boff(std::integral_constant<int, ks>{}));
                constexpr bool LA0 = n0 + ZS_D < ZS_NA, LA1 = DUAL && n1 + ZS_D < ZS_NA;
                if constexpr (LA0) A[n0 % ZS_D] = a_loadn(std::integral_constant<int, LA0 ? n0 + ZS_D : 0>{});
                if constexpr (LA1) A[n1 % ZS_D] = a_loadn(std::integral_constant<int, LA1 ? n1 + ZS_D : 0>{});
                constexpr bool NXT = tap == 4 && (ks < 3 || HASNEXT);
                const f16x_t zero = {};
                g_for<0, 4>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    if constexpr (WHICH != 1) {
                        if constexpr (t == 0 && NB != 40) g_mma_c(accz[p], av0, B[tap + p], zero);
                        else g_mma(accz[p], av0, B[tap + p]);
                    }
                    if constexpr (WHICH != 0) {
                        if constexpr (t == 0 && NB != 40) g_mma_c(accq[p], av1, B[tap + p], zero);
                        else g_mma(accq[p], av1, B[tap + p]);
                    }
                    if constexpr (NXT) B[p] = *(const uint4*)((ks < 3 ? tb : tbn) + p * G_LS + boff(std::integral_constant<int, (ks + 1) & 3>{}));
                });
                // issue order: MFMA, LDS read, MFMA, weight load, ...: the loads of a step go out behind its first MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (tap < 4 || NXT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (LA0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if constexpr (NXT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (LA1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if constexpr (NXT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (NXT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if constexpr (DUAL) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        constexpr float KS_SIG = -1.4426950408889634f, KS_TANH = 2.8853900817779268f;
        typedef float f2_t __attribute__((ext_vector_type(2)));
        const char* const tbH = S0 + lh * 4 * G_LS;
        const char* const tbR = S1 + lh * 4 * G_LS;
        const char* const tbX = S2 + lh * 4 * G_LS;
        const float* const SB = (const float*)(smem + 3 * PIECE);
        int it = it0;
        [[maybe_unused]] int trk = 0;
        prime();
        G_STAMP(wave, trk, 0);
        asm volatile("s_barrier" ::: "memory");                 // A0: the first tile's h, r*h have landed; the bias is in LDS
        for (;;) {
            int img, P0, L0;
            pos_of(it, img, P0, L0);
            G_STAMP(wave, trk, 1);
            b_init(tbH);
            __builtin_amdgcn_sched_barrier(0);
            phase(tbH, tbR, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::true_type{});
            phase(tbR, tbR, std::integral_constant<int, 20>{}, std::integral_constant<int, 1>{}, std::false_type{});
            G_STAMP(wave, trk, 2);
            asm volatile("s_barrier" ::: "memory");             // B: x has landed; h, r*h are free
            G_STAMP(wave, trk, 3);
            b_init(tbX);
            __builtin_amdgcn_sched_barrier(0);
            phase(tbX, tbX, std::integral_constant<int, 40>{}, std::integral_constant<int, 2>{}, std::false_type{});
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(accz[0]), "v"(accz[1]), "v"(accz[2]), "v"(accz[3]), "v"(accq[0]), "v"(accq[1]), "v"(accq[2]), "v"(accq[3]));
#endif
            G_STAMP(wave, trk, 4);
            asm volatile("s_barrier" ::: "memory");             // C: the x piece is free (the staged h' goes there)
            G_STAMP(wave, trk, 5);
            // h of this wave's outputs: clamped addresses, unconditional loads, all eight in flight under the first pass of transcendentals
            const char* hb = a.h.ptr + (int64_t)img * a.h.sb + (int64_t)min(P0 + lx, a.Plen - 1) * a.h.sp + (cs * 32 + hi * 8) * 2;
            u4_t hreg[4][2];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const char* hp = hb + (int64_t)min(L0 + lh * 4 + p, a.Llen - 1) * a.h.sl;
                hreg[p][0] = *gcp<u4_t>(hp);
                hreg[p][1] = *gcp<u4_t>(hp + 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // bias (packed cout order == MFMA row order) folded into the exponent's packed fma
                f2_t bz[8], bq[8];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f4_t vz = *(const f4_t*)(SB + cs * 32 + g * 8 + hi * 4);
                    const f4_t vq = *(const f4_t*)(SB + 64 + cs * 32 + g * 8 + hi * 4);
                    bz[2 * g] = f2_t{vz[0], vz[1]} * KS_SIG;  bz[2 * g + 1] = f2_t{vz[2], vz[3]} * KS_SIG;
                    bq[2 * g] = f2_t{vq[0], vq[1]} * KS_TANH; bq[2 * g + 1] = f2_t{vq[2], vq[3]} * KS_TANH;
                }
                // pass 1 (no h needed): a = e^-z', b = e^2q' (exponent clamped: b stays finite), den = (b + 1)(1 + a); r = 1 / den -> accz, b -> accq
                g_for<0, 4>([&](auto P) {
                    constexpr int p = decltype(P)::value;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        f2_t ez = f2_t{accz[p][2 * i], accz[p][2 * i + 1]} * KS_SIG + bz[i];
                        f2_t eq = f2_t{accq[p][2 * i], accq[p][2 * i + 1]} * KS_TANH + bq[i];
                        eq = f2_t{__builtin_fminf(eq.x, 60.0f), __builtin_fminf(eq.y, 60.0f)};
                        const f2_t av = f2_t{__builtin_amdgcn_exp2f(ez.x), __builtin_amdgcn_exp2f(ez.y)} + 1.0f;
                        const f2_t bv = f2_t{__builtin_amdgcn_exp2f(eq.x), __builtin_amdgcn_exp2f(eq.y)};
                        const f2_t den = (bv + 1.0f) * av;
                        accz[p][2 * i] = __builtin_amdgcn_rcpf(den.x);
                        accz[p][2 * i + 1] = __builtin_amdgcn_rcpf(den.y);
                        accq[p][2 * i] = bv.x;
                        accq[p][2 * i + 1] = bv.y;
                    }
                });
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::"v"(accz[0]), "v"(accz[1]), "v"(accz[2]), "v"(accz[3]), "v"(accq[0]), "v"(accq[1]), "v"(accq[2]), "v"(accq[3]));
#endif
                __builtin_amdgcn_sched_barrier(0);               // ALL of pass 1 first (interleaved with pass 2, the first h load's latency was exposed)
                G_STAMP(wave, trk, 6);
#if DEMFI_ZQS_PRIME_EARLY
                prime();                                         // the next tile's first A fragments: in flight under pass 2
                __builtin_amdgcn_sched_barrier(0);
#endif
                // pass 2: h' = h + ((b - 1) - h (b + 1)) r, fp16, staged in the x piece (line lh 4 + p) for the helpers' whole-line stores
                g_for<0, 4>([&](auto P) {
                    constexpr int p = decltype(P)::value;
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) {
                        const u4_t rr = hreg[p][m2];
                        u4_t o = {0u, 0u, 0u, 0u};
#pragma unroll
                        for (int d = 0; d < 4; ++d) {            // dword d = channels 2 d, 2 d + 1 of the lane's 8 = accumulator elements 8 m2 + 2 d, + 1
                            const int i0 = 8 * m2 + 2 * d;
                            const f2_t bv = f2_t{accq[p][i0], accq[p][i0 + 1]};
                            const f2_t bp = bv + 1.0f, bm = bv - 1.0f;
                            const float nlo = g_nfma_lo(bp.x, rr[d], bm.x);      // (b - 1) - h (b + 1)
                            const float nhi = g_nfma_hi(bp.y, rr[d], bm.y);
                            unsigned pk = 0u;
                            g_fma16_lo(pk, nlo, accz[p][i0], rr[d]);
                            g_fma16_hi(pk, nhi, accz[p][i0 + 1], rr[d]);
                            o[d] = pk;
                        }
                        *(u4_t*)(S2 + (lh * 4 + p) * G_LS + soff[m2]) = o;
                    }
                });
            }
            G_STAMP(wave, trk, 7);
#if !DEMFI_ZQS_PRIME_EARLY
            prime();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G_STAMP(wave, trk, 8);
            asm volatile("s_barrier" ::: "memory");             // E: h' is staged; the next tile's h, r*h have landed
            ++trk;
            ++it;
            if (it >= it1) break;
        }
        return;
    }

    // ================= MFMA waves ================================================================================
    const int hi = lane >> 5, lx = lane & 31;
    const int cs = wave & 1, role = wave >> 1;                   // cout half; ZQ: 0 = z, 1 = q;  R: line half
    const unsigned lane16 = lane * 16;
    // weights of this wave: chunk 0 / chunk 1 in the order the phases walk them
    const char* const wA = (MODE == GM_ZQ ? (role ? a.w1[0] : a.w0[0]) : a.w0[1]) + cs * 1024;     // phase A: ZQ h / r*h; R: x
    const char* const wB = (MODE == GM_ZQ ? (role ? a.w1[1] : a.w0[1]) : a.w0[0]) + cs * 1024;     // phase B: ZQ x;       R: h
    const float* const bias = (MODE == GM_ZQ && role) ? a.b1 : a.b0;
    const int boff0 = lx * 128 + ((hi ^ ((lx >> 1) & 7)) << 4);  // B fragment of k-step ks: boff0 ^ (ks << 5) (swizzle touches bits 4-6 only)
    auto boff = [&](auto KS) {
        int b = boff0;
        asm volatile("" : "+v"(b));
        return b ^ (decltype(KS)::value << 5);
    };
    int soff[2];                                                 // staged / exchanged 16-byte piece of this lane: channels cs*32 + m2*16 + hi*8 ..
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) soff[m2] = lx * 128 + (((cs * 4 + m2 * 2 + hi) ^ ((lx >> 1) & 7)) << 4);
    auto a_load = [&](const char* w, auto T) {
        constexpr int t = decltype(T)::value, ks = t / 5, tap = t % 5;
        const char* wb = w;
        asm volatile("" : "+s"(wb));                             // uniform base + 32-bit lane offset: the saddr form, nothing hoisted
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));
        return __builtin_bit_cast(uint4, *gcp<u4_t>(wb + (unsigned)(((tap * 4 + ks) * 2) * 1024 + l16)));
    };
    uint4 A[G_DEPTH];
    f16x_t acc[8];
    g_for<0, G_DEPTH>([&](auto T) { A[decltype(T)::value] = a_load(wA, T); });

    // one 64-channel piece: 4 k-steps x 5 taps of 8 MFMAs.  tb = first window line of this wave.  B window: step (ks, tap)
    // multiplies output line p with window line tap + p; line 8 + tap arrives during tap (for tap + 1); during tap 4 the next
    // k-step's lines 0..7 are read, each behind the MFMA that uses line 4 + p for the last time: 9-10 fragments live.
    auto conv_phase = [&](const char* tb, const char* wcur, const char* wnxt, auto INITC, const f16x_t& cinit, auto LAST) {
        uint4 B[12];
        {
            const int o = boff(std::integral_constant<int, 0>{});
            g_for<0, 8>([&](auto R) { B[decltype(R)::value] = *(const uint4*)(tb + decltype(R)::value * G_LS + o); });
        }
        __builtin_amdgcn_sched_barrier(0);
        g_for<0, G_NSTEP>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            constexpr int ks = t / 5, tap = t % 5;
            const uint4 av = A[t % G_DEPTH];
            if constexpr (tap < 4) B[8 + tap] = *(const uint4*)(tb + (8 + tap) * G_LS + boff(std::integral_constant<int, ks>{}));
            // LAST: the ring is re-primed after the epilogue instead (16 registers the epilogue needs)
            if constexpr (t + G_DEPTH < G_NSTEP) A[t % G_DEPTH] = a_load(wcur, std::integral_constant<int, t + G_DEPTH>{});
            else if constexpr (!decltype(LAST)::value) A[t % G_DEPTH] = a_load(wnxt, std::integral_constant<int, t + G_DEPTH - G_NSTEP>{});
            if constexpr (tap < 4 || ks == 3) {
                g_for<0, 8>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    if constexpr (decltype(INITC)::value && t == 0) g_mma_c(acc[p], av, B[tap + p], cinit);
                    else g_mma(acc[p], av, B[tap + p]);
                });
                constexpr bool has_a = t + G_DEPTH < G_NSTEP || !decltype(LAST)::value;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (tap < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (has_a) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            } else {
                uint4 Bn[8];
                const int o = boff(std::integral_constant<int, (ks + 1) & 3>{});
                g_for<0, 8>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    g_mma(acc[p], av, B[4 + p]);
                    Bn[p] = *(const uint4*)(tb + p * G_LS + o);
                });
                g_for<0, 8>([&](auto P) { B[decltype(P)::value] = Bn[decltype(P)::value]; });
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

    constexpr float KS_SIG = -1.4426950408889634f, KS_TANH = 2.8853900817779268f;
    typedef float f2_t __attribute__((ext_vector_type(2)));
    // bias as the C operand of every accumulator's first MFMA (packed order == MFMA row order); loaded before the barrier in front of phase A
    auto load_bias = [&](f16x_t& c1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f4_t q = *gcp<f4_t>(bias + cs * 32 + g * 8 + hi * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) c1[g * 4 + j] = q[j];
        }
    };
    int it = it0;
    [[maybe_unused]] int trk = 0;
    f16x_t c1;
    load_bias(c1);
    if constexpr (MODE == GM_ZQ) {
        G_STAMP(wave, trk, 0);
        asm volatile("s_barrier" ::: "memory");                 // A0: the first tile's h, r*h have landed (later tiles: barrier E says so)
    }
    for (;;) {
        int img, P0, L0;
        pos_of(it, img, P0, L0);
        if constexpr (MODE == GM_R) load_bias(c1);
        if constexpr (MODE == GM_ZQ) {
            const bool is_q = role != 0;
            G_STAMP(wave, trk, 1);
            conv_phase(is_q ? S1 : S0, wA, wB, std::true_type{}, c1, std::false_type{});
            G_STAMP(wave, trk, 2);
            asm volatile("s_barrier" ::: "memory");             // B: x has landed; h, r*h are free
            G_STAMP(wave, trk, 3);
            {
                const f16x_t none = {};
                conv_phase(S2, wB, wA, std::false_type{}, none, std::true_type{});
            }
#if defined(DEMFI_TRACE) && defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
#endif
            G_STAMP(wave, trk, 4);
            asm volatile("s_barrier" ::: "memory");             // C: the x piece is free
            G_STAMP(wave, trk, 5);
            // two separate paths from here to the end of the tile (barriers D, E inside both): merged, the allocator carried a second copy
            // of the accumulators through the other path and spilled.  The multiplies / adds are packed fp32 (v_pk_*: two values per issue
            // slot; a lone wave's epilogue is priced by its instruction count), the transcendentals cannot be.
            if (is_q) {
                // q~ = tanh(.) = 1 - 2 / (1 + e^(2 .)), 16 values at a time (the transcendentals' latency needs the ILP), handed to the z waves
                // as fp16 through the x piece
                g_for<0, 8>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    f2_t v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        f2_t e = f2_t{acc[p][2 * i], acc[p][2 * i + 1]} * KS_TANH;
                        e = f2_t{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} + 1.0f;
                        e = f2_t{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
                        v[i] = e * -2.0f + 1.0f;
                    }
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) {
                        h8_t o;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {           // quads 2 m2, 2 m2 + 1 = pairs 4 m2 .. 4 m2 + 3
                            o[2 * j] = (half_t)v[4 * m2 + j].x;
                            o[2 * j + 1] = (half_t)v[4 * m2 + j].y;
                            o[4 + 2 * j] = (half_t)v[4 * m2 + 2 + j].x;
                            o[4 + 2 * j + 1] = (half_t)v[4 * m2 + 2 + j].y;
                        }
                        *(u4_t*)(S2 + p * G_LS + soff[m2]) = __builtin_bit_cast(u4_t, o);
                    }
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                G_STAMP(wave, trk, 6);
                asm volatile("s_barrier" ::: "memory");         // D: q~ is in LDS
                G_STAMP(wave, trk, 7);
                g_for<0, G_DEPTH>([&](auto T) { A[decltype(T)::value] = a_load(wA, T); });
                load_bias(c1);
                G_STAMP(wave, trk, 8);
                asm volatile("s_barrier" ::: "memory");         // E: h' is staged; the next tile's h, r*h have landed
            } else {
                // z waves: z = sigmoid(.) stays in the accumulators (fp32); then the blend h' = h + z (q~ - h) for this wave's values.
                // h: clamped addresses, unconditional loads, two lines per batch, three batches in flight
                const char* hb = a.h.ptr + (int64_t)img * a.h.sb + (int64_t)min(P0 + lx, a.Plen - 1) * a.h.sp + (cs * 32 + hi * 8) * 2;
                u4_t hreg[3][2][2];
                auto h_fetch = [&](auto PB) {
                    constexpr int pb = decltype(PB)::value;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const char* hp = hb + (int64_t)min(L0 + pb + p, a.Llen - 1) * a.h.sl;
                        hreg[(pb >> 1) % 3][p][0] = *gcp<u4_t>(hp);
                        hreg[(pb >> 1) % 3][p][1] = *gcp<u4_t>(hp + 32);
                    }
                };
                h_fetch(std::integral_constant<int, 0>{});
                h_fetch(std::integral_constant<int, 2>{});
                h_fetch(std::integral_constant<int, 4>{});
                __builtin_amdgcn_sched_barrier(0);               // the loads go out BEFORE the sigmoids (the scheduler sank them to the barrier)
                g_for<0, 8>([&](auto P) {
                    constexpr int p = decltype(P)::value;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        f2_t e = f2_t{acc[p][2 * i], acc[p][2 * i + 1]} * KS_SIG;
                        e = f2_t{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} + 1.0f;
                        acc[p][2 * i] = __builtin_amdgcn_rcpf(e.x);
                        acc[p][2 * i + 1] = __builtin_amdgcn_rcpf(e.y);
                    }
                });
#if defined(__HIP_DEVICE_COMPILE__)
                // the sigmoids run HERE, beside the q waves' tanh (pure register arithmetic would otherwise sink below the barrier)
                asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
#endif
                G_STAMP(wave, trk, 6);
                asm volatile("s_barrier" ::: "memory");         // D: q~ is in LDS
                G_STAMP(wave, trk, 7);
                g_for<0, 4>([&](auto PP) {
                    constexpr int pb = 2 * decltype(PP)::value;
                    // both lines' q~ first (four LDS reads in flight), then the differences, then the packed fmas: independent chains side by side
                    u4_t qq[2][2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) qq[p][m2] = *(const u4_t*)(S2 + (pb + p) * G_LS + soff[m2]);
                    }
                    g_for<0, 2>([&](auto P) {
                        constexpr int p = pb + decltype(P)::value;
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) {
                            const u4_t rr = hreg[(pb >> 1) % 3][p - pb][m2];
                            const u4_t qv = qq[p - pb][m2];
                            float dl[8];
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                dl[2 * d] = g_sub16_lo(qv[d], rr[d]);
                                dl[2 * d + 1] = g_sub16_hi(qv[d], rr[d]);
                            }
                            u4_t o = {0u, 0u, 0u, 0u};
#pragma unroll
                            for (int d = 0; d < 4; ++d) {       // dword d = channels 2 d, 2 d + 1 of the lane's 8: quad 2 m2 + (d >> 1), elements 2 (d & 1), + 1
                                const int i0 = (2 * m2 + (d >> 1)) * 4 + 2 * (d & 1);
                                unsigned pk = 0u;
                                g_fma16_lo(pk, acc[p][i0], dl[2 * d], rr[d]);
                                g_fma16_hi(pk, acc[p][i0 + 1], dl[2 * d + 1], rr[d]);
                                o[d] = pk;
                            }
                            *(u4_t*)(S2 + p * G_LS + soff[m2]) = o;
                        }
                    });
                    if constexpr (pb + 6 < 8) h_fetch(std::integral_constant<int, pb + 6>{});
                });
                g_for<0, G_DEPTH>([&](auto T) { A[decltype(T)::value] = a_load(wA, T); });
                load_bias(c1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                G_STAMP(wave, trk, 8);
                asm volatile("s_barrier" ::: "memory");         // E: h' is staged; the next tile's h, r*h have landed
            }
        } else {
            // R: phase A contracts x (piece 0), phase B h (piece 1); r * h from the h window's centre lines
            const char* const t0 = S0 + role * 8 * G_LS;
            const char* const t1 = S1 + role * 8 * G_LS;
            G_STAMP(wave, trk, 0);
            asm volatile("s_barrier" ::: "memory");             // A
            G_STAMP(wave, trk, 1);
            conv_phase(t0, wA, wB, std::true_type{}, c1, std::false_type{});
            G_STAMP(wave, trk, 2);
            asm volatile("s_barrier" ::: "memory");             // B
            G_STAMP(wave, trk, 3);
            {
                const f16x_t none = {};
                conv_phase(t1, wB, wA, std::false_type{}, none, std::false_type{});
            }
#if defined(DEMFI_TRACE) && defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
#endif
            G_STAMP(wave, trk, 4);
            // r * h: h from the window's centre lines; barrier C (the h piece is free) AFTER the epilogue.  Measured alternative (round 6): h
            // into registers first and C before the epilogue, so that the helpers issue the next h under it -- the tile got 20 % SLOWER
            // (epilogue + stores 5 500 -> 11 200 cycles: the helpers' DMA and these 16 partial-line stores share the CU's memory pipe)
            char* const ob = a.dst + (int64_t)img * a.d_sb + (int64_t)(P0 + lx) * a.d_sp + (cs * 32 + hi * 8) * 2;
            const bool pok = P0 + lx < a.Plen;
            g_for<0, 8>([&](auto P) {
                constexpr int p = decltype(P)::value;
                const int l = L0 + role * 8 + p;
                f2_t sg[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {                    // sigmoid, 16 values at a time, packed multiplies / adds
                    f2_t e = f2_t{acc[p][2 * i], acc[p][2 * i + 1]} * KS_SIG;
                    e = f2_t{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} + 1.0f;
                    sg[i] = f2_t{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
                }
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const h8_t r = __builtin_bit_cast(h8_t, *(const u4_t*)(t1 + (2 + p) * G_LS + soff[m2]));
                    h8_t o;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {               // quads 2 m2, 2 m2 + 1 = pairs 4 m2 .. 4 m2 + 3
                        o[2 * j] = (half_t)__builtin_fmaf(sg[4 * m2 + j].x, (float)r[2 * j], 0.0f);
                        o[2 * j + 1] = (half_t)__builtin_fmaf(sg[4 * m2 + j].y, (float)r[2 * j + 1], 0.0f);
                        o[4 + 2 * j] = (half_t)__builtin_fmaf(sg[4 * m2 + 2 + j].x, (float)r[4 + 2 * j], 0.0f);
                        o[4 + 2 * j + 1] = (half_t)__builtin_fmaf(sg[4 * m2 + 2 + j].y, (float)r[4 + 2 * j + 1], 0.0f);
                    }
                    if (pok && l < a.Llen) *gp<u4_t>(ob + (int64_t)l * a.d_sl + m2 * 32) = __builtin_bit_cast(u4_t, o);
                }
            });
            G_STAMP(wave, trk, 5);
            asm volatile("s_barrier" ::: "memory");             // C: the h piece is free
        }
        ++trk;
        ++it;
        if (it >= it1) break;
    }
}

// a 1x5 / 5x1 fp16 layer over two 64-channel NHWC pieces with 64 outputs in the SepConvGRU packing (what conv_sep5_c128_persist_kernel
// takes at cout_pad 64): the building block both launches are made of
bool g_layer_ok(const demfi_conv* h)
{
    if (h->dtype != DEMFI_F16 || h->stride != 1 || !h->zero_page || !h->wpack || !h->bias || !h->cout_perm) return false;
    if (!((h->kh == 1 && h->kw == 5) || (h->kh == 5 && h->kw == 1))) return false;
    if (h->pad_y != h->kh / 2 || h->pad_x != h->kw / 2 || h->inH != h->H || h->inW != h->W) return false;
    if (h->n_chunks != 2 || h->cout_pad != 64 || h->nco != 2 || h->n_segs < 1) return false;
    for (int c = 0; c < 2; ++c) {
        const demfi_chunk& ch = h->chunks[c];
        if (ch.n_pieces != 1 || ch.nks != 4) return false;
        const demfi_piece& p = h->pieces[ch.first_piece];
        if (!p.fat || p.nch != 64 || p.up_shift || !p.v.ptr || p.v.sc != 1 || p.v.is_f32) return false;
        if (p.v.sy * 2 * 24 + p.v.sx * 2 * 40 >= (int64_t)1 << 31) return false;        // 32-bit per-lane offsets inside a window
    }
    const int sgi = h->sub_seg[0];
    if (sgi < 0 || h->sub_seg[1] != sgi) return false;
    for (int o = 0; o < 8; ++o)
        if (h->oct_seg[o] != sgi || h->oct_n[o] != 8 || h->oct_ch[o] != h->oct_ch[0] + 8 * o) return false;
    const demfi_seg& sg = h->segs[sgi];
    if (sg.scale != 1 || sg.dy || sg.dx || !sg.dst.ptr || sg.dst.is_f32 || sg.dst.sc != 1) return false;
    if (h->pack.ptr || h->u8_sink) return false;
    return true;
}
bool g_same_view(const demfi_view& a, const demfi_view& b)
{
    return a.ptr == b.ptr && a.sx == b.sx && a.sy == b.sy && a.sb == b.sb && a.sc == b.sc && a.is_f32 == b.is_f32;
}
GPiece g_piece(const demfi_view& v, bool tr)
{
    return GPiece{(const char*)v.ptr, (tr ? v.sy : v.sx) * 2, (tr ? v.sx : v.sy) * 2, v.sb * 2};
}
const char* g_wchunk(const demfi_conv* h, int c) { return (const char*)h->wpack + h->chunks[c].w_off * 16; }

template <int MODE>
int g_launch(GArgs& a, const demfi_conv* h, const demfi_seg& out, void* stream)
{
    const bool tr = h->kh == 5;
    a.dst = (char*)out.dst.ptr + (int64_t)h->oct_ch[0] * 2;
    a.d_sl = (tr ? out.dst.sy : out.dst.sx) * 2; a.d_sp = (tr ? out.dst.sx : out.dst.sy) * 2; a.d_sb = out.dst.sb * 2;
    a.zeros = (const char*)h->zero_page;
    a.Llen = tr ? h->H : h->W; a.Plen = tr ? h->W : h->H; a.batch = h->batch;
    a.n_pt = (a.Plen + 31) / 32;
    a.n_ls = (a.Llen + GCfg<MODE>::TL - 1) / GCfg<MODE>::TL;
    const int64_t total = (int64_t)a.n_pt * a.n_ls * a.batch;
    if (total <= 0 || total >= (int64_t)1 << 30) return demfi_set_error(DEMFI_ERR_ARG, "demfi_gru: empty or oversized launch");
    if (a.d_sl * 24 + a.d_sp * 40 >= (int64_t)1 << 31) return demfi_set_error(DEMFI_ERR_ARG, "demfi_gru: destination strides exceed 32-bit lane offsets");
    DEMFI_LDS_ATTR(gru_sep5_kernel<MODE>);
    const int grid = total >= 256 ? 256 : (int)total;
    hipLaunchKernelGGL(gru_sep5_kernel<MODE>, dim3(grid), dim3(G_NTHREADS), GCfg<MODE>::LDS, (hipStream_t)stream, a);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}

}  // namespace

// r * h of a SepConvGRU half-step: hr = the 64-cout reset-gate layer over [h, x] with the MUL epilogue (res == h)
extern "C" int demfi_gru_r_eligible(const demfi_conv* hr)
{
    if (!hr || !g_layer_ok(hr)) return 0;
    const demfi_seg& sg = hr->segs[hr->sub_seg[0]];
    if (sg.mode != DEMFI_MODE_MUL || !sg.res.ptr) return 0;
    const demfi_view& h = hr->pieces[hr->chunks[0].first_piece].v;
    demfi_view res = sg.res;
    res.ptr = (char*)res.ptr + (int64_t)hr->oct_ch[0] * 2;
    if (!g_same_view(res, h)) return 0;                          // the h window in LDS is the residual
    if (sg.dst.ptr == h.ptr) return 0;
    return 1;
}

extern "C" int demfi_gru_r(const demfi_conv* hr, void* stream)
{
    if (!demfi_gru_r_eligible(hr))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_gru_r: not the reset-gate layer of a SepConvGRU half-step (1x5 / 5x1 fp16, [h, x] -> 64, MUL epilogue with res == h)");
    const bool tr = hr->kh == 5;
    GArgs a = {};
    a.h = g_piece(hr->pieces[hr->chunks[0].first_piece].v, tr);
    a.x = g_piece(hr->pieces[hr->chunks[1].first_piece].v, tr);
    a.rh = a.h;
    a.w0[0] = g_wchunk(hr, 0); a.w0[1] = g_wchunk(hr, 1);
    a.w1[0] = a.w0[0]; a.w1[1] = a.w0[1];
    a.b0 = a.b1 = hr->bias;
    return g_launch<GM_R>(a, hr, hr->segs[hr->sub_seg[0]], stream);
}

// z, q and the state update of a half-step in one launch: hz = update-gate layer over [h, x] (sigmoid, STORE -> the z buffer, which the
// fused kernel never touches), hq = candidate layer over [r*h, x] with the GRU epilogue (res == h, aux == hz's destination)
extern "C" int demfi_gru_zq_eligible(const demfi_conv* hz, const demfi_conv* hq)
{
    if (!hz || !hq || !g_layer_ok(hz) || !g_layer_ok(hq)) return 0;
    if (hz->kh != hq->kh || hz->kw != hq->kw || hz->H != hq->H || hz->W != hq->W || hz->batch != hq->batch) return 0;
    const demfi_seg& sz = hz->segs[hz->sub_seg[0]];
    const demfi_seg& sq = hq->segs[hq->sub_seg[0]];
    if (sz.mode != DEMFI_MODE_STORE || sz.act != DEMFI_ACT_SIGMOID || sz.res.ptr) return 0;
    if (sq.mode != DEMFI_MODE_GRU || !sq.res.ptr || !sq.aux.ptr) return 0;
    const demfi_view& h = hz->pieces[hz->chunks[0].first_piece].v;
    const demfi_view& xz = hz->pieces[hz->chunks[1].first_piece].v;
    const demfi_view& xq = hq->pieces[hq->chunks[1].first_piece].v;
    if (!g_same_view(xz, xq)) return 0;
    demfi_view res = sq.res, aux = sq.aux, zd = sz.dst;
    res.ptr = (char*)res.ptr + (int64_t)hq->oct_ch[0] * 2;
    aux.ptr = (char*)aux.ptr + (int64_t)hq->oct_ch[0] * 2;
    zd.ptr = (char*)zd.ptr + (int64_t)hz->oct_ch[0] * 2;
    if (!g_same_view(res, h) || !g_same_view(aux, zd)) return 0;
    if (sq.dst.ptr == h.ptr) return 0;                           // not in place: neighbouring tiles read the halo lines
    return 1;
}

extern "C" int demfi_gru_zq(const demfi_conv* hz, const demfi_conv* hq, void* stream)
{
    if (!demfi_gru_zq_eligible(hz, hq))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_gru_zq: not the (update gate, candidate) pair of a SepConvGRU half-step (1x5 / 5x1 fp16, [h, x] -> 64 "
                                              "sigmoid and [r*h, x] -> 64 with the GRU epilogue on the same h, x, z)");
    const bool tr = hz->kh == 5;
    GArgs a = {};
    a.h = g_piece(hz->pieces[hz->chunks[0].first_piece].v, tr);
    a.x = g_piece(hz->pieces[hz->chunks[1].first_piece].v, tr);
    a.rh = g_piece(hq->pieces[hq->chunks[0].first_piece].v, tr);
    a.w0[0] = g_wchunk(hz, 0); a.w0[1] = g_wchunk(hz, 1);
    a.w1[0] = g_wchunk(hq, 0); a.w1[1] = g_wchunk(hq, 1);
    a.b0 = hz->bias; a.b1 = hq->bias;
    // DEMFI_GRU_ZQS=0: the round-6a form (z waves and q waves, q~ through LDS)
    static const bool zqs = [] { const char* e = getenv("DEMFI_GRU_ZQS"); return !(e && e[0] == '0'); }();
    return zqs ? g_launch<GM_ZQS>(a, hq, hq->segs[hq->sub_seg[0]], stream) : g_launch<GM_ZQ>(a, hq, hq->segs[hq->sub_seg[0]], stream);
}

#ifdef DEMFI_TRACE
extern "C" int demfi_gru_trace_dump(unsigned long long* out, int64_t n)
{
    const int64_t have = (int64_t)GT_WGS * GT_WAVES * GT_TILES * GT_STAMPS;
    if (n != have) return demfi_set_error(DEMFI_ERR_ARG, "demfi_gru_trace_dump: expected %lld entries", (long long)have);
    DEMFI_HIP_CHECK(hipDeviceSynchronize());
    DEMFI_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gru_trace), have * 8));
    static const std::vector<unsigned long long> zeros(have, 0ull);
    DEMFI_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_gru_trace), zeros.data(), have * 8));
    return DEMFI_OK;
}
#endif
