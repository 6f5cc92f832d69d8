// The streamed-weight kernel of round 3 (Ch_Reducer) and its 3x3 / 32-cout instantiation of round 5 (a unit of its own since round 6; the kernel is
// unchanged).  wsconv.hip is the round-6 family with helper-wave DMA.
#include "conv_common.h"

namespace {

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


}  // namespace

DEMFI_TU_TRACE(demfi_wstream_trace_collect)

bool demfi_wstream_eligible(const demfi_conv* h, int ks, int nch) { return wstream_eligible(h, ks, nch); }
bool demfi_wstream3_on() { return wstream3_on(); }
int demfi_wstream_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st) { return launch_wstream(h, dev, st); }
int demfi_wstream3_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st) { return launch_wstream3(h, dev, st); }
