// The SepConvGRU kernel of rounds 1-5 (DEMFI_GRU6=0; a unit of its own since round 6; the kernel is unchanged).  Round 6: gru.hip.
#include "conv_common.h"

namespace {

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


}  // namespace

DEMFI_TU_TRACE(demfi_sep_trace_collect)

bool demfi_sep_eligible(const demfi_conv* h) { return sep_eligible(h); }

int demfi_sep_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, bool* fall_through)
{
    *fall_through = false;
#ifdef DEMFI_ABLATION
    static const int svar = getenv("DEMFI_SEP_VARIANT") ? atoi(getenv("DEMFI_SEP_VARIANT")) : 0;
    if (svar == 1) return launch_sep<1>(h, dev, st);
    if (svar == 2) return launch_sep<2>(h, dev, st);
    if (svar == 3) return launch_sep<3>(h, dev, st);
    if (svar == 4) return launch_sep<4>(h, dev, st);
    if (svar == -1) { *fall_through = true; return DEMFI_OK; }
#endif
    return launch_sep(h, dev, st);
}
