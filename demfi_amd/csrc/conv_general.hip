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
#include "conv_common.h"

namespace {

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

int demfi_conv_general_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, size_t lds)
{
    return h->dtype == DEMFI_F16 ? dispatch<half_t>(h, dev, st, lds) : dispatch<float>(h, dev, st, lds);
}
