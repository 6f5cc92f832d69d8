// Streamed-weight convolution, second family (round 6): 64 (or 32) output channels per work item over ANY number of 32-channel input units,
//   * 3x3 stride 1 (+ optional NHWC residual; units may be read through the x2 nearest-neighbour upsample of the UNet decoder,
//     DeMFInet.py:592-601)                           -- layers whose K is not one 64-channel record: dec0 / dec1 / dec2, FGAC's w_gen;
//   * 4x4 stride 2 as FOUR PHASES of 2x2 taps        -- the UNet encoders (Refine_Module.enc1/2/3, DeMFInet.py:575-577, 588-590);
//   * NCH = 1: 32-cout blocks, four row groups of a 32 x 32-pixel tile -- the 48 RDB growth convolutions of the trunk (DeMFInet.py:266-281).
// What it replaces: the general kernel (conv_kernel) ran the stride-2 layers at 0.09-0.16 of the matrix peak -- one workgroup per
// 8 x 32 tile, a VGPR-staged gather of a 18 x 66-pixel window per chunk and one barrier per tap (16 per chunk).
//
// Stride 2 by phases: output (y, x) = sum_{ky, kx < 4} W[ky][kx] . In(2y - 1 + ky, 2x - 1 + kx).  Input rows of parity py are touched by
// the taps ky = 1 - py and 3 - py only, i.e. the phase image P_py,px(r, q) = In(2r + py, 2q + px) -- a view with doubled strides -- sees a
// 2 x 2 stride-1 filter whose window starts at r = y - py.  A (32-channel chunk, phase) pair is one UNIT: its haloed tile
// (17 lines x 33 records of 64 bytes) goes to LDS by DMA, its 2 x 2 x 2 k-steps are 8 steps of 8 MFMAs per wave.  No tap is multiplied
// by zero (the embedded-3x3 form would do 2.25x the work) and nothing is gathered through registers.
//
// Structure (resblock.hip's roles): MFMA waves 0-3 = cout half x row half, eight 32x32 accumulators each (8 rows x 32 pixels x 32
// couts); their only VMEM is the A fragment of a step -- one global_load_dwordx4 from the packed, L2-resident weights into a register
// ring (saddr form) -- so the compiler's vmcnt bookkeeping is exact.  Helper waves 4-7 issue the unit DMA NBUF - 1 units ahead into a
// ring of NBUF unit buffers (3x3: 3 x 40 KiB, 2x2: 4 x 36 KiB: a 2x2 unit is only 2 048 matrix-pipe cycles, less than the HBM latency)
// and count their own vmcnt.  ONE s_barrier per unit: "unit u has landed" and "the buffer of unit u - 1 is free".
// Inside a unit the steps are (kx, k-step, ky) with ky innermost and a rolling window of B lines (line l serves output row p at
// ky = l - p): per step 8 MFMAs, one or two ds_read_b128, one A load.
// Epilogue: the accumulators START at bias + residual (the residual of the NEXT item is fetched in front of this item's stores), so
// the end of an item is activation, conversion and 16-byte stores of 8 consecutive channels per lane (cout_perm packing).
// What bounds it (profiles/r06_notes.md section 6, DEMFI_WS2_TRACE): the helpers' DMA runs at the CU's share of the memory system (~10 B/clk: the
// 2x2 and the 32-cout forms are bound by it, the 3x3 / 64-cout form is balanced); the item's end (64 KiB of the MFMA waves' own stores [+ loads]) is
// serial with the matrix phases -- staged whole-line stores by the helpers, lane-swapped store layouts and a one-instruction DMA address path
// were each built and measured neutral.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include <string.h>

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void ws_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ws_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ void ws_mma(f16x_t& acc, const uint4& a, const uint4& b)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), acc, 0, 0, 0);
}

template <bool S2, int NCH = 2> struct Ws2Cfg {                // NCH: 32-cout subtiles per work item (2: waves = cout half x row half; 1: four row groups)
    static constexpr int KSE = S2 ? 2 : 3;                       // taps per dimension a unit sees
    static constexpr int RPW = 8;                                // rows per MFMA wave
    static constexpr int TH = RPW * (4 / NCH), TWP = 32;         // output tile: 16 x 32 (NCH 2) or 32 x 32 (NCH 1)
    static constexpr int LH = TH + KSE - 1, LL = TWP + KSE - 1;  // lines / records per line of a unit's tile
    static constexpr int NHW = 4;                                // helper (DMA) waves
    static constexpr int NI = (LH * LL + 15) / 16;               // DMA instructions per unit (16 records of 64 bytes each)
    static constexpr int NIW = (NI + NHW - 1) / NHW;             // per helper wave: every helper issues exactly NIW (the surplus ones land in the buffer's pad)
    static constexpr int BUF_BYTES = NIW * NHW * 1024;
    static constexpr int NBUF = S2 ? 4 : (NCH == 2 ? 3 : 2);
    static constexpr int LDS_BYTES = NBUF * BUF_BYTES + 1024;    // + the biases of up to four cout blocks
    static constexpr int NG = 2 * KSE;                           // (kx, k-step) groups per unit
    static constexpr int NSTEP = NG * KSE;
#ifndef DEMFI_WS2_DEPTH3
#define DEMFI_WS2_DEPTH3 6
#endif
    static constexpr int DEPTH = S2 ? 4 : DEMFI_WS2_DEPTH3;      // A prefetch distance in steps
    static constexpr int BL = KSE + RPW - 1;                     // lines of a wave's B window
    static constexpr int NTHREADS = 256 + 64 * NHW;
    static constexpr int CB = 64 * NCH;                          // bytes of a block's channels in an output record
    static_assert(NCH == 1 || NCH == 2, "one or two 32-cout subtiles");
    static_assert(!(S2 && NCH == 1), "the 2x2 phase form is built for 64-cout blocks");
    static_assert(NSTEP % DEPTH == 0, "static ring indices");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((NBUF - 1) * NIW <= 63, "a helper's units in flight must be countable in vmcnt");
};

// everything a wave needs to know about one unit; all wave-uniform
struct WsUnit {
    const char* src;            // address of record (line 0, column 0) of the unit's tile (may lie outside the buffer: such records are never loaded)
    const char* w;              // A fragments of this unit, cout half 0: + (step constant + cs) KiB + lane * 16
    int iy0, ix0;               // input coordinates of record (0, 0)
    int sx, sy;                 // byte strides between the tile's records / lines
    int half;                   // 1: only the first 16 channels of the unit are real (the rest reads the zero page)
    const char* src2;           // half: an optional second piece of 8 channels (one 16-byte slot) behind the first 16
    int sx2, sy2, has2;
    int up;                     // 1: the piece is read through a nearest-neighbour x2 upsample (3x3 only): src = the image, record (l, c) = pixel ((iy0 + l) >> 1, (ix0 + c) >> 1)
    bool interior;
};

// The launch description is a KERNEL ARGUMENT (by value), not the device copy of the descriptor: kernarg memory is invariant to the
// compiler, so nothing is re-read behind the kernel's barriers, and a unit's description is one scalar load instead of the dependent
// chain chunk -> piece -> view (the first version walked the descriptor: ~13 us of scalar-load latency chains per item).
constexpr int WS_MAX_CHUNKS = 16, WS_MAX_BLOCKS = 4;
struct WsChunk {
    const char* src; const char* src2;      // piece base (image 0); second piece of a tail unit (or NULL)
    int64_t sb, sb2;                        // bytes between images
    int sx, sy, sx2, sy2;                   // bytes between pixels / rows (of the piece itself)
    int flags;                              // 1: tail unit (16 real channels [+ 8 of src2]), 2: read through the x2 upsample
    int w_off;                              // bytes: this chunk's packed weights inside a cout block
};
struct WsBlock {
    char* dst; const char* res;             // first of the block's 64 channels
    int64_t d_sb, r_sb;
    int d_sx, d_sy, r_sx, r_sy;             // bytes
    int relu, _pad;
};
struct WsArgs {
    const char* wpack; const float* bias; const char* zeros;
    int64_t w_blk_stride;                   // bytes between cout blocks
    int H, W, inH, inW, batch, n_chunks, nblk, _pad;
    unsigned long long* trace;              // phase stamps of workgroup 0 (DEMFI_WS2_TRACE=1, a debugging aid; NULL in the product)
    WsBlock blk[WS_MAX_BLOCKS];
    WsChunk ch[WS_MAX_CHUNKS];
};

// (fp16 half of a packed pair) * 1.0 + c in one VALU op: bias + residual -> the accumulator's initial value
__device__ __forceinline__ float ws_mix_lo(unsigned a, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}
__device__ __forceinline__ float ws_mix_hi(unsigned a, float c)
{
    float d = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
#endif
    return d;
}

constexpr int WS_NTHREADS = 512;                                 // 4 MFMA + 4 helper waves (== Ws2Cfg::NTHREADS: a macro argument cannot hold the template's comma)
template <bool S2, int NCH>
__global__ __launch_bounds__(WS_NTHREADS, 1) void conv_ws2_kernel(const WsArgs a)
{
    using C = Ws2Cfg<S2, NCH>;
    static_assert(C::NTHREADS == WS_NTHREADS, "launch bounds");
    constexpr int KSE = C::KSE, RPW = C::RPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, inH = a.inH, inW = a.inW;
    const int tiles_x = (W + C::TWP - 1) / C::TWP, tiles_y = (H + C::TH - 1) / C::TH, tiles_img = tiles_x * tiles_y;
    const int nblk = a.nblk;                                     // cout blocks (32 NCH couts): the innermost index of an item (they share the input tile)
    const int total = tiles_img * a.batch * nblk;
    // contiguous run of items per workgroup, the workgroups of an XCD (blockIdx % 8) share a contiguous band
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
    const int upi = a.n_chunks * (S2 ? 4 : 1);                   // units per item
    const int n_units = (it1 - it0) * upi;
    const char* const zeros = a.zeros;
    int tr_n = 0;
    auto stamp = [&](int tag) {                                  // [wave][512] of (tag << 56 | s_memtime), workgroup 0 only
        if (a.trace && blockIdx.x == 0 && tr_n < 512 && lane == 0) a.trace[wave * 512 + tr_n] = ((unsigned long long)tag << 56) | (__builtin_readcyclecounter() & 0xffffffffffffffull);
        ++tr_n;
    };

    auto item_pos = [&](int it, int& blk, int& img, int& ty, int& tx) {
        blk = it % nblk;
        const int tl = it / nblk;
        img = tl / tiles_img;
        const int rem = tl - img * tiles_img;
        ty = rem / tiles_x;
        tx = rem - ty * tiles_x;
    };
    // walks the units of this workgroup in order; the divisions of item_pos happen once per item
    struct Cursor { int it, ph, cu, blk, img, ty, tx; };
    auto cursor_at = [&](int it) {
        Cursor c;
        c.it = it; c.ph = 0; c.cu = 0;
        item_pos(it, c.blk, c.img, c.ty, c.tx);
        return c;
    };
    auto advance = [&](Cursor& c) {
        if (++c.cu < a.n_chunks) return;                        // chunk innermost: the two halves of a 128-byte line follow each other
        c.cu = 0;
        if (S2 && ++c.ph < 4) return;
        c.ph = 0;
        if (c.it + 1 < it1) { ++c.it; item_pos(c.it, c.blk, c.img, c.ty, c.tx); }      // past the last unit: stays on the last item (never used for data)
    };
    auto unit_info = [&](const Cursor& c) {
        WsUnit r;
        const int py = c.ph >> 1, px = c.ph & 1;
        const WsChunk& ch = a.ch[c.cu];
        const int st = S2 ? 2 : 1;
        r.sx = ch.sx * st;
        r.sy = ch.sy * st;
        r.iy0 = S2 ? 2 * c.ty * C::TH - py : c.ty * C::TH - 1;
        r.ix0 = S2 ? 2 * c.tx * C::TWP - px : c.tx * C::TWP - 1;
        r.up = S2 ? 0 : (ch.flags >> 1) & 1;
        r.src = ch.src + c.img * ch.sb;
        if (!r.up) r.src += (int64_t)r.iy0 * ch.sy + (int64_t)r.ix0 * ch.sx;
        r.half = ch.flags & 1;
        r.has2 = ch.src2 != nullptr;
        r.sx2 = ch.sx2 * st;
        r.sy2 = ch.sy2 * st;
        r.src2 = ch.src2 + c.img * ch.sb2 + (int64_t)r.iy0 * ch.sy2 + (int64_t)r.ix0 * ch.sx2;      // only used when has2
        // packed weights: [cout block][chunk][tap][k-step][cout half] KiB; S2: tap (ky, kx) = (2 ky2 + 1 - py, 2 kx2 + 1 - px)
        r.w = a.wpack + c.blk * a.w_blk_stride + ch.w_off + (S2 ? ((4 * (1 - py) + (1 - px)) * 4) * 1024 : 0);
        const int ly = S2 ? 2 * (C::LH - 1) : C::LH - 1, lx_ = S2 ? 2 * (C::LL - 1) : C::LL - 1;
        r.interior = r.iy0 >= 0 && r.iy0 + ly < inH && r.ix0 >= 0 && r.ix0 + lx_ < inW;
        return r;
    };

    if (wave >= 4) {
        // ================= helper waves: the unit DMA, NBUF - 1 units ahead ==============================================
        const int dw = wave - 4;
        __builtin_assume(dw >= 0 && dw < C::NHW);
        // (measured, round 6: the 2x2 form is bound by the helpers' DMA issue -- 9 instructions of ~400 cycles per 2 048-cycle unit; s_setprio 3
        // on the helpers changes nothing: enc1#t 0.400 / 0.401 ms same box)
        // instruction i = dw + 4 j covers records 16 i .. 16 i + 15; lane -> (record 16 i + lane / 4, physical slot lane % 4), logical slot
        // (8 channels) = physical ^ ((column >> 2) & 3)
        int dl[C::NIW], dc[C::NIW], dslot[C::NIW];
#pragma unroll
        for (int j = 0; j < C::NIW; ++j) {
            const int rec = min((dw + C::NHW * j) * 16 + (lane >> 2), C::LH * C::LL - 1);      // lanes past the tile re-read its last record
            const int l = rec / C::LL, c = rec - l * C::LL;
            dl[j] = l; dc[j] = c;
            dslot[j] = (lane & 3) ^ ((c >> 2) & 3);
        }
        Cursor dcur = cursor_at(it0);
        auto issue_unit = [&](int u) {
            const WsUnit un = unit_info(dcur);
            advance(dcur);
            char* const buf = smem + (u % C::NBUF) * C::BUF_BYTES;
#pragma unroll
            for (int j = 0; j < C::NIW; ++j) {
                const int i = dw + C::NHW * j;
                const int ll = un.up ? (un.iy0 + dl[j]) >> 1 : dl[j], cc = un.up ? (un.ix0 + dc[j]) >> 1 : dc[j];      // (un.up: wave-uniform)
                const char* g = un.src + (ll * un.sy + cc * un.sx + (dslot[j] << 4));
                bool ok = !((un.half != 0) & (dslot[j] >= 2));
                if (un.has2) {                                   // wave-uniform: slot 2 of a tail unit comes from the second piece
                    const bool s2nd = dslot[j] == 2;
                    g = s2nd ? un.src2 + (dl[j] * un.sy2 + dc[j] * un.sx2) : g;
                    ok = ok | s2nd;
                }
                if (!un.interior) {                              // wave-uniform
                    const int iy = un.iy0 + dl[j] * (S2 ? 2 : 1), ix = un.ix0 + dc[j] * (S2 ? 2 : 1);
                    ok = ok & ((unsigned)iy < (unsigned)inH) & ((unsigned)ix < (unsigned)inW);
                }
                if (!ok) g = zeros;
                __builtin_amdgcn_global_load_lds((const DEMFI_GLOBAL void*)g, (__attribute__((address_space(3))) void*)(buf + i * 1024), 16, 0, 0);
            }
        };
        const int pre = min(C::NBUF - 1, n_units);
        for (int u = 0; u < pre; ++u) issue_unit(u);
        for (int u = 0; u < n_units; ++u) {
            // units u .. min(u + NBUF - 2, n_units - 1) are in flight; unit u must have landed
            const int behind = min(C::NBUF - 2, n_units - 1 - u);
            if (behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::NIW) : "memory");
            else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NIW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(1);
            asm volatile("s_barrier" ::: "memory");             // unit u is in LDS; the MFMA waves are done with unit u - 1
            stamp(2);
            if (u + C::NBUF - 1 < n_units) issue_unit(u + C::NBUF - 1);      // into the buffer of unit u - 1
            stamp(3);
        }
        return;
    }

    // ================= MFMA waves ====================================================================================
    const int hi = lane >> 5, lx = lane & 31;
    const int cs = NCH == 2 ? (wave & 1) : 0, rh = NCH == 2 ? (wave >> 1) : wave;      // cout half, row group (RPW rows each)
    const unsigned lane16 = lane * 16;
    // B fragment of (group g = (kx, k-step), line l): record (lx + kx) of line rh * 8 + l, 16-byte slot (2 ksl + hi) swizzled by the column
    int boff[C::NG];
#pragma unroll
    for (int g = 0; g < C::NG; ++g) {
        const int col = lx + (g >> 1);
        boff[g] = (rh * RPW * C::LL + col) * 64 + ((((g & 1) * 2 + hi) ^ ((col >> 2) & 3)) << 4);
    }
    auto a_load = [&](const WsUnit& un, auto T) {
        constexpr int t = decltype(T)::value;
        constexpr int g = t / KSE, ky = t - g * KSE, kx = g >> 1, ksl = g & 1;
        constexpr int tap = S2 ? 8 * ky + 2 * kx : ky * 3 + kx;  // S2: + the phase's (4 (1 - py) + (1 - px)), folded into un.w
        const char* wb = un.w + cs * 1024;
        asm volatile("" : "+s"(wb));                             // opaque uniform base + one 32-bit lane offset: the saddr form, nothing hoisted
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));
        return __builtin_bit_cast(uint4, *gcp<u4_t>(wb + (unsigned)(((tap * 2 + ksl) * NCH) * 1024 + l16)));
    };

    f16x_t acc[RPW];
    uint4 A[C::DEPTH];
    // biases of all cout blocks to LDS once (behind the unit buffers); an item's four quads per lane are four ds_read_b128
    float* const bias_lds = (float*)(smem + C::LDS_BYTES - 1024);
    if (wave == 0)
        for (int i = lane; i < nblk * 32 * NCH; i += 64) bias_lds[i] = a.bias[i];
    // accumulators of an item start at bias + residual
    auto res_fetch = [&](const Cursor& io, u4_t (&rr)[RPW][2]) {
        const WsBlock& bk = a.blk[io.blk];
        if (!bk.res) return;                                     // wave-uniform; acc_init does not read rr then
        const int ox = io.tx * C::TWP + lx;
        const int oy0 = io.ty * C::TH + rh * RPW;
        // uniform base (image, first row, first channel) + ONE 32-bit lane offset: the saddr form, no per-load 64-bit address registers
        const char* const rb = bk.res + io.img * bk.r_sb + (int64_t)oy0 * bk.r_sy + cs * 64;
        const unsigned loff = (unsigned)(ox * bk.r_sx + hi * 16);
#pragma unroll
        for (int p = 0; p < RPW; ++p) {
            // pixels outside the image read the zero page: every load is unconditional, all sixteen are in flight together
            const bool ok = oy0 + p < H && ox < W;
            const char* base = ok ? rb + p * bk.r_sy : zeros;   // (row: uniform)
            const unsigned lo_ = ok ? loff : 0u;
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) rr[p][m2] = *gcp<u4_t>(base + (lo_ + (ok ? m2 * 32 : 0)));
        }
    };
    auto acc_init = [&](const Cursor& io, const u4_t (&rr)[RPW][2], auto FIRST) {
        f4_t bq[4];                                              // bias in MFMA-row order: quads 0..3 of this lane
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            if constexpr (decltype(FIRST)::value) bq[qd] = *gcp<f4_t>(a.bias + io.blk * 32 * NCH + cs * 32 + qd * 8 + hi * 4);      // no barrier yet: from global
            else bq[qd] = *(const f4_t*)(bias_lds + io.blk * 32 * NCH + cs * 32 + qd * 8 + hi * 4);
        }
        if (a.blk[io.blk].res) {
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const u4_t r = rr[p][m2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[p][(2 * m2) * 4 + 2 * q] = ws_mix_lo(r[q], bq[2 * m2][2 * q]);
                        acc[p][(2 * m2) * 4 + 2 * q + 1] = ws_mix_hi(r[q], bq[2 * m2][2 * q + 1]);
                        acc[p][(2 * m2 + 1) * 4 + 2 * q] = ws_mix_lo(r[2 + q], bq[2 * m2 + 1][2 * q]);
                        acc[p][(2 * m2 + 1) * 4 + 2 * q + 1] = ws_mix_hi(r[2 + q], bq[2 * m2 + 1][2 * q + 1]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[p][qd * 4 + j] = bq[qd][j];
                }
            }
        }
    };
    auto item_store = [&](const Cursor& io) {
        const WsBlock& bk = a.blk[io.blk];
        const int ox = io.tx * C::TWP + lx;
        const int oy0 = io.ty * C::TH + rh * RPW;
        char* const ob = bk.dst + io.img * bk.d_sb + (int64_t)oy0 * bk.d_sy + cs * 64;      // uniform
        const unsigned loff = (unsigned)(ox * bk.d_sx + hi * 16);
        const h8_t z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int p = 0; p < RPW; ++p) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                h8_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = (half_t)acc[p][(2 * m2) * 4 + j];
                    o[4 + j] = (half_t)acc[p][(2 * m2 + 1) * 4 + j];
                }
                if (bk.relu) o = __builtin_elementwise_max(o, z8);     // wave-uniform; on the packed halves
                if (oy0 + p < H && ox < W) *gp<h8_t>(ob + p * bk.d_sy + (loff + m2 * 32)) = o;
            }
        }
    };

    Cursor ccur = cursor_at(it0);                               // the unit on the matrix cores
    Cursor io = ccur;                                            // its item (blk, img, ty, tx)
    {
        u4_t rr[RPW][2];
        res_fetch(io, rr);
        acc_init(io, rr, std::true_type{});
    }
    WsUnit cur = unit_info(ccur);
    ws_for<0, C::DEPTH>([&](auto T) { A[decltype(T)::value] = a_load(cur, T); });

    for (int u = 0; u < n_units; ++u) {
        const bool last_of_item = ccur.cu == a.n_chunks - 1 && (!S2 || ccur.ph == 3);
        advance(ccur);                                           // past the end: stays on the last item, its A fragments are loaded and dropped
        const WsUnit nxt = unit_info(ccur);
        const char* const tb = smem + (u % C::NBUF) * C::BUF_BYTES;
        stamp(1);
        asm volatile("s_barrier" ::: "memory");                 // unit u has landed
        stamp(2);
        uint4 B[C::BL];
        ws_for<0, RPW>([&](auto R) { B[decltype(R)::value] = *(const uint4*)(tb + boff[0] + decltype(R)::value * (C::LL * 64)); });
        __builtin_amdgcn_sched_barrier(0);
        ws_for<0, C::NSTEP>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            constexpr int g = t / KSE, ky = t % KSE;
            const uint4 av = A[t % C::DEPTH];
            // lines RPW .. BL - 1 of a group arrive during its steps 0 .. KSE - 2; the NEXT group's lines 0 .. RPW - 1 are read during its
            // last step, each right behind the MFMA that uses line KSE - 1 + p for the last time
            if constexpr (ky < KSE - 1) B[ky + RPW] = *(const uint4*)(tb + boff[g] + (ky + RPW) * (C::LL * 64));
            if constexpr (t + C::DEPTH < C::NSTEP) A[t % C::DEPTH] = a_load(cur, std::integral_constant<int, t + C::DEPTH>{});
            else                                   A[t % C::DEPTH] = a_load(nxt, std::integral_constant<int, t + C::DEPTH - C::NSTEP>{});
            if constexpr (ky < KSE - 1 || g + 1 == C::NG) {
                ws_for<0, RPW>([&](auto P) { ws_mma(acc[decltype(P)::value], av, B[ky + decltype(P)::value]); });
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (ky < KSE - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            } else {
                uint4 Bn[RPW];
                ws_for<0, RPW>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    ws_mma(acc[p], av, B[KSE - 1 + p]);
                    Bn[p] = *(const uint4*)(tb + boff[g + 1 < C::NG ? g + 1 : 0] + p * (C::LL * 64));
                });
                ws_for<0, RPW>([&](auto P) { B[decltype(P)::value] = Bn[decltype(P)::value]; });
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
        stamp(3);
        if (last_of_item) {
            // ---- end of an item: the next item's residual is requested in front of this item's stores
            const bool more = u + 1 < n_units;                  // then ccur already stands on the next item's first unit
            u4_t rr[RPW][2];
            if (more) res_fetch(ccur, rr);
            stamp(4);
            item_store(io);
            stamp(5);
            if (more) acc_init(ccur, rr, std::false_type{});
            io = ccur;
            stamp(6);
        }
        cur = nxt;
    }
}

static bool ws2_on()
{
    static const bool on = !(getenv("DEMFI_WS2") && atoi(getenv("DEMFI_WS2")) == 0);      // A/B: 0 = these layers stay on the kernels of rounds 1-5
    return on;
}

}  // namespace

// 3x3 stride 1 / 4x4 stride 2, fp16, units of 32 channels (64-byte records; a unit = one NHWC piece of 32 channels, or of 16 + zero
// padding), 64-cout blocks each routed to 64 consecutive channels of one NHWC fp16 destination (optional NHWC fp16 residual)
bool demfi_ws2_eligible(const demfi_conv* h)
{
    if (!ws2_on() || h->dtype != DEMFI_F16 || !h->zero_page || h->rec_bytes != 64 || (h->nco != 2 && h->nco != 1) || h->cout_pad % (32 * h->nco)) return false;
    const int nco = h->nco, nblk = h->cout_pad / (32 * nco);
    const bool s2 = h->stride == 2;
    if (nco == 1) {
        // 32-cout blocks (the RDB growth convolutions, DeMFInet.py:266-281): 3x3 only; DEMFI_WS2_RDB=0 leaves them on the round-5 kernel
        static const bool rdb = !(getenv("DEMFI_WS2_RDB") && atoi(getenv("DEMFI_WS2_RDB")) == 0);
        if (!rdb || s2) return false;
    }
    if (s2) {
        if (h->kh != 4 || h->kw != 4 || h->pad_y != 1 || h->pad_x != 1 || h->inH != 2 * h->H || h->inW != 2 * h->W) return false;
    } else {
        if (h->stride != 1 || h->kh != 3 || h->kw != 3 || h->pad_y != 1 || h->pad_x != 1 || h->inH != h->H || h->inW != h->W) return false;
    }
    if (h->n_chunks < (nco == 1 ? 2 : 1) || h->n_chunks > WS_MAX_CHUNKS || nblk > WS_MAX_BLOCKS) return false;
    if (h->w_blk_stride * 16 * nblk >= (int64_t)1 << 31) return false;
    for (int c = 0; c < h->n_chunks; ++c) {
        const demfi_chunk& ch = h->chunks[c];
        if (ch.nks != 2 || ch.n_pieces < 1 || ch.n_pieces > 3) return false;
        const demfi_piece& p = h->pieces[ch.first_piece];
        if (!p.fat || !p.v.ptr || p.v.sc != 1 || p.v.is_f32 || p.lds_ch != 0) return false;
        if (p.up_shift != 0 && (p.up_shift != 1 || s2 || (h->H & 1) || (h->W & 1))) return false;      // x2 nearest-neighbour upsample: 3x3 only
        if (ch.n_pieces == 1) {
            if (p.nch != 32) return false;
        } else {
            // a tail unit: 16 channels [+ 8 channels of a second NHWC piece] + zero padding
            const demfi_piece& q = h->pieces[ch.first_piece + 1];
            if (p.nch != 16 || p.up_shift) return false;
            if (ch.n_pieces == 2) {
                if (q.v.ptr != nullptr || q.nch != 16) return false;
            } else {
                const demfi_piece& z = h->pieces[ch.first_piece + 2];
                if (!q.v.ptr || !q.fat || q.nch != 8 || q.up_shift || q.v.sc != 1 || q.v.is_f32 || z.v.ptr != nullptr || z.nch != 8) return false;
                if (q.v.sy * 4 * 40 + q.v.sx * 4 * 40 >= (int64_t)1 << 31) return false;
            }
        }
        if (p.v.sy * 4 * 40 + p.v.sx * 4 * 40 >= (int64_t)1 << 31) return false;       // 32-bit per-lane offsets inside a tile
    }
    for (int b = 0; b < nblk; ++b) {
        const int sgi = h->sub_seg[nco * b];
        if (sgi < 0 || (nco == 2 && h->sub_seg[2 * b + 1] != sgi)) return false;
        for (int o = 0; o < 4 * nco; ++o)
            if (h->oct_seg[4 * nco * b + o] != sgi || h->oct_n[4 * nco * b + o] != 8 || h->oct_ch[4 * nco * b + o] != h->oct_ch[4 * nco * b] + 8 * o) return false;
        const demfi_seg& sg = h->segs[sgi];
        if (sg.mode != DEMFI_MODE_STORE || sg.scale != 1 || sg.dy || sg.dx || !sg.dst.ptr || sg.dst.is_f32 || sg.dst.sc != 1 || sg.aux.ptr) return false;
        if (sg.act != DEMFI_ACT_NONE && sg.act != DEMFI_ACT_RELU) return false;
        if (sg.res.ptr && (sg.res.is_f32 || sg.res.sc != 1)) return false;
    }
    if (h->u8_sink || h->pack.ptr) return false;
    return true;
}

int demfi_ws2_launch(const demfi_conv* h, const demfi_conv* /*dev*/, void* stream)
{
    const int nco = h->nco, th = nco == 2 ? 16 : 32;
    const int64_t total = (int64_t)((h->W + 31) / 32) * ((h->H + th - 1) / th) * h->batch * (h->cout_pad / (32 * nco));
    if (total <= 0 || total >= (int64_t)1 << 30) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d (streamed-weight 64-cout kernel): empty or oversized launch");
    if (!h->wpack || !h->bias) return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv2d (streamed-weight 64-cout kernel): weights / bias not bound");
    WsArgs a;
    memset(&a, 0, sizeof(a));
    a.wpack = (const char*)h->wpack; a.bias = h->bias; a.zeros = (const char*)h->zero_page;
    a.w_blk_stride = h->w_blk_stride * 16;
    a.H = h->H; a.W = h->W; a.inH = h->inH; a.inW = h->inW; a.batch = h->batch; a.n_chunks = h->n_chunks; a.nblk = h->cout_pad / (32 * nco);
    for (int b = 0; b < a.nblk; ++b) {
        const demfi_seg& sg = h->segs[h->sub_seg[nco * b]];
        const int ch0 = h->oct_ch[4 * nco * b];
        WsBlock& k = a.blk[b];
        k.dst = (char*)sg.dst.ptr + (int64_t)ch0 * 2; k.d_sb = sg.dst.sb * 2; k.d_sx = (int)(sg.dst.sx * 2); k.d_sy = (int)(sg.dst.sy * 2);
        k.res = sg.res.ptr ? (const char*)sg.res.ptr + (int64_t)ch0 * 2 : nullptr;
        k.r_sb = sg.res.sb * 2; k.r_sx = (int)(sg.res.sx * 2); k.r_sy = (int)(sg.res.sy * 2);
        k.relu = sg.act == DEMFI_ACT_RELU;
    }
    for (int c = 0; c < h->n_chunks; ++c) {
        const demfi_chunk& ch = h->chunks[c];
        const demfi_piece& p = h->pieces[ch.first_piece];
        WsChunk& k = a.ch[c];
        k.src = (const char*)p.v.ptr; k.sb = p.v.sb * 2; k.sx = (int)(p.v.sx * 2); k.sy = (int)(p.v.sy * 2);
        k.flags = (p.nch == 16 ? 1 : 0) | (p.up_shift ? 2 : 0);
        k.w_off = (int)(ch.w_off * 16);
        if (ch.n_pieces == 3) {
            const demfi_piece& q = h->pieces[ch.first_piece + 1];
            k.src2 = (const char*)q.v.ptr; k.sb2 = q.v.sb * 2; k.sx2 = (int)(q.v.sx * 2); k.sy2 = (int)(q.v.sy * 2);
        }
    }
    const int grid = total >= 256 ? 256 : (int)total;
    static const bool tracing = getenv("DEMFI_WS2_TRACE") && atoi(getenv("DEMFI_WS2_TRACE")) != 0;
    static unsigned long long* trbuf = nullptr;
    if (tracing) {
        if (!trbuf) { DEMFI_HIP_CHECK(hipMalloc(&trbuf, 8 * 512 * 8)); }
        DEMFI_HIP_CHECK(hipMemset(trbuf, 0, 8 * 512 * 8));
        a.trace = trbuf;
    }
    constexpr int lds_s2 = Ws2Cfg<true, 2>::LDS_BYTES, lds_3 = Ws2Cfg<false, 2>::LDS_BYTES, lds_31 = Ws2Cfg<false, 1>::LDS_BYTES;
    if (h->stride == 2) {
        DEMFI_LDS_ATTR((conv_ws2_kernel<true, 2>));
        hipLaunchKernelGGL((conv_ws2_kernel<true, 2>), dim3(grid), dim3(WS_NTHREADS), lds_s2, (hipStream_t)stream, a);
    } else if (nco == 2) {
        DEMFI_LDS_ATTR((conv_ws2_kernel<false, 2>));
        hipLaunchKernelGGL((conv_ws2_kernel<false, 2>), dim3(grid), dim3(WS_NTHREADS), lds_3, (hipStream_t)stream, a);
    } else {
        DEMFI_LDS_ATTR((conv_ws2_kernel<false, 1>));
        hipLaunchKernelGGL((conv_ws2_kernel<false, 1>), dim3(grid), dim3(WS_NTHREADS), lds_31, (hipStream_t)stream, a);
    }
    DEMFI_HIP_CHECK(hipGetLastError());
    if (tracing) {
        // debugging aid: the stamps of workgroup 0 (MFMA wave 0, helper wave 4) as deltas in cycles, one line per stamp
        static unsigned long long host[8 * 512];
        DEMFI_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        DEMFI_HIP_CHECK(hipMemcpy(host, trbuf, sizeof(host), hipMemcpyDeviceToHost));
        for (int w : {0, 4}) {
            fprintf(stderr, "[ws2 trace] stride %d chunks %d blocks %d wave %d:", h->stride, h->n_chunks, h->cout_pad / (32 * nco), w);
            unsigned long long prev = 0;
            for (int i = 0; i < 512 && host[w * 512 + i]; ++i) {
                const unsigned long long t = host[w * 512 + i] & 0xffffffffffffffull;
                fprintf(stderr, " %d:%lld", (int)(host[w * 512 + i] >> 56), prev ? (long long)(t - prev) : 0ll);
                prev = t;
            }
            fprintf(stderr, "\n");
        }
    }
    return DEMFI_OK;
}
