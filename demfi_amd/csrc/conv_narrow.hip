// The narrow persistent kernel (a unit of its own since round 6; the kernel is unchanged).
#include "conv_common.h"

namespace {

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
// P7 (7x7 only, round 6): PAIRED TAPS.  Mixer.conv_delta1 feeds 5 real channels: the upper 8 of a tap's 16-channel k-step multiply zeros.  When the
// chunk is [8-channel piece | 8 zero channels], the upper-half lanes of the B operand read the NEXT column's first 8 channels instead, and the A
// fragment of the step is assembled (in the weight copy to LDS) from the lower halves of the taps (ky, 2j) and (ky, 2j + 1): one MFMA = two taps,
// 28 k-steps instead of 49; with ky innermost the 8 input rows of a column pair serve both output rows at all 7 ky (15 LDS reads per 14 MFMAs
// instead of 21).
template <int NCO, int REC, int EPI, int KS = 3, int NDMA = NarrowCfg<REC, KS>::NDMA, int NOCT = 4, bool PACK = false, bool P7 = false>
__global__ __launch_bounds__(NT + 64 * NDMA, 1) void conv3x3_narrow_persist_kernel(const demfi_conv* __restrict__ d)
{
    static_assert(!P7 || (KS == 7 && NCO == 1 && REC == 32), "paired taps: the 7x7 / 16-channel / 32-cout instantiation");
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
        if constexpr (P7) {
            // paired fragment (ky, j): lanes 0-31 = channels 0-7 of tap (ky, 2j), lanes 32-63 = channels 0-7 of tap (ky, 2j + 1) (kx = 7: zeros)
            for (int i = dw; i < 7 * 4; i += NDMA) {
                const int ky = i >> 2, j = i & 3;
                const int kx = 2 * j + (lane >> 5);
                const void* g = kx < 7 ? (const void*)(wsrc + (ky * 7 + kx) * 64 + (lane & 31)) : (const void*)zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(wlds + i * 1024), 16, 0, 0);
            }
        } else
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
        } else if constexpr (P7) {
            struct PairFrag { uint4 a[7]; uint4 b[8]; };
            auto load_j = [&](PairFrag& f, auto J_) {
                constexpr int j = decltype(J_)::value;
                const int col = lx + 2 * j + hi;                 // upper-half lanes: the next column's first 8 channels
                const char* p0 = tb + col * REC + ((0 ^ Cfg::swz(col)) << 4);
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) f.a[ky] = *(const uint4*)(wl + (ky * 4 + j) * 1024);
#pragma unroll
                for (int r = 0; r < 8; ++r) f.b[r] = *(const uint4*)(p0 + r * (P_LW * REC));
            };
            PairFrag f[2];
            load_j(f[0], std::integral_constant<int, 0>{});
            static_for<0, 4>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                if constexpr (j + 1 < 4) load_j(f[(j + 1) & 1], std::integral_constant<int, j + 1>{});
                static_for<0, 7>([&](auto KY_) {
                    constexpr int ky = decltype(KY_)::value;
                    Mma<half_t>::run(acc[0][0], f[j & 1].a[ky], f[j & 1].b[ky]);
                    Mma<half_t>::run(acc[0][1], f[j & 1].a[ky], f[j & 1].b[ky + 1]);
                });
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // the next pair's 15 reads between this pair's 14 MFMAs
                if constexpr (j + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int q = 1; q < 14; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (j + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
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
    else {
        if constexpr (KS == 7 && NCO == 1 && REC == 32) {
            // paired taps (P7): the chunk is one 8-channel NHWC piece + 8 zero channels; DEMFI_N7_PAIR=0: one tap per k-step (rounds 2-5)
            static const bool pair_on = !(getenv("DEMFI_N7_PAIR") && atoi(getenv("DEMFI_N7_PAIR")) == 0);
            const demfi_chunk& ch = h->chunks[0];
            if (pair_on && ch.n_pieces == 2 && h->pieces[ch.first_piece].nch == 8 && h->pieces[ch.first_piece].v.ptr && h->pieces[ch.first_piece].lds_ch == 0 &&
                h->pieces[ch.first_piece + 1].v.ptr == nullptr) {
                // four DMA waves: with the matrix phase at 2 400 cycles the two of the unpaired form (4 000 cycles for their 9 instructions each:
                // per-lane piece / bounds arithmetic beside an MFMA wave) would set the period
                constexpr int ND = 4;
                DEMFI_LDS_ATTR((conv3x3_narrow_persist_kernel<1, 32, 0, 7, ND, 4, false, true>));
                hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<1, 32, 0, 7, ND, 4, false, true>), dim3(grid), dim3(NT + 64 * ND), lds, st, dev);
                DEMFI_HIP_CHECK(hipGetLastError());
                return DEMFI_OK;
            }
        }
        hipLaunchKernelGGL((conv3x3_narrow_persist_kernel<NCO, REC, 0, KS>), dim3(grid), dim3(NT + 64 * NarrowCfg<REC, KS>::NDMA), lds, st, dev);
    }
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}


}  // namespace

DEMFI_TU_KNOB(demfi_narrow_set_knob)
DEMFI_TU_TRACE(demfi_narrow_trace_collect)

int demfi_narrow_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, bool thin, bool* handled)
{
    *handled = true;
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
    *handled = false;
    return DEMFI_OK;
}
