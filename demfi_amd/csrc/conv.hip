// demfi_conv2d: which kernel owns a convolution descriptor (replaces every nn.Conv2d / nn.Conv3d(1,k,k) call site of the reference together with the
// torch.cat / PixelShuffle / UpsamplingNearest2d / activation / residual / GRU-gate ops around them: DeMFInet.py:209-231, 324-378, 575-584, 30-44, 800-868;
// SURVEY.md section 2.2 C1, C6-C12).  The kernels live in their own translation units (until round 6 this was one 3 100-line file):
//   conv_general.hip  the general implicit-GEMM kernel (any shape, fp16 / fp32)           conv_c64.hip   3x3 over one 64-channel piece (32 couts; 64: staged stores)
//   conv_narrow.hip   3x3 / 7x7 over <= 64 channels in one chunk, NHWC or thin outputs      conv_sep.hip   SepConvGRU 1x5 / 5x1 (rounds 1-5; round 6: gru.hip)
//   conv_wstream.hip  Ch_Reducer 7x7 with streamed weights (+ its 3x3 / 32-cout form)       wsconv.hip     32-channel units, helper-wave DMA (round 6)
//   resblock.hip      fused residual block                                                  gru.hip        SepConvGRU half-step as r*h, then z + q + blend
// Eligibility is decided from the descriptor alone; demfi_conv_build (ctx.cpp) shapes a layer for the kernel that owns it.
#include "conv_common.h"

namespace {

bool sep_eligible(const demfi_conv* h) { return demfi_sep_eligible(h); }
bool wstream_eligible(const demfi_conv* h, int ks = 7, int nch = 2) { return demfi_wstream_eligible(h, ks, nch); }
bool wstream3_on() { return demfi_wstream3_on(); }

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
            if (k) { demfi_c64_set_knob(k); demfi_narrow_set_knob(k); }
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
    bool fall = false, handled = false;
    if (sep_eligible(h)) {
        const int rc = demfi_sep_launch(h, dev, st, &fall);
        if (!fall) return rc;
    }
    if (wstream_eligible(h)) return demfi_wstream_launch(h, dev, st);
    if (!persist_eligible(h) && !narrow_eligible(h) && demfi_ws2_eligible(h)) return demfi_ws2_launch(h, dev, st);      // wsconv.hip (round 6)
    if (wstream3_on() && wstream_eligible(h, 3, 1)) return demfi_wstream3_launch(h, dev, st);
    if (persist_eligible(h)) {
        const int rc = demfi_c64_launch(h, dev, st, &fall);
        if (!fall) return rc;
    } else if (narrow_eligible(h)) {
#ifdef DEMFI_ABLATION
        if (!(getenv("DEMFI_NARROW_OFF") && atoi(getenv("DEMFI_NARROW_OFF"))))
#endif
        {
            const int rc = demfi_narrow_launch(h, dev, st, !persist_out_eligible(h), &handled);
            if (handled) return rc;
        }
    }
    return demfi_conv_general_launch(h, dev, st, (size_t)lds);
}

#ifdef DEMFI_TRACE
// trace build only: copy the phase trace out (see TRACE_STAMP; one kernel family is traced at a time, the units' buffers are OR-ed) and clear it
extern "C" int demfi_trace_dump(unsigned long long* out, int64_t n)
{
    const int64_t have = (int64_t)TR_WGS * TR_WAVES * TR_TILES * TR_STAMPS;
    if (!out || n < have) return demfi_set_error(DEMFI_ERR_ARG, "demfi_trace_dump: need room for %lld entries", (long long)have);
    DEMFI_HIP_CHECK(hipDeviceSynchronize());
    for (int64_t i = 0; i < have; ++i) out[i] = 0;
    int st = demfi_c64_trace_collect(out);
    if (st >= 0) st = demfi_narrow_trace_collect(out);
    if (st >= 0) st = demfi_sep_trace_collect(out);
    if (st >= 0) st = demfi_wstream_trace_collect(out);
    return st < 0 ? st : (int)have;
}
#endif
