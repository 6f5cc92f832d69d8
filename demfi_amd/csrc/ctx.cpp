// Forward context of libdemfi_hip.so: the launch plan of the DeMFI-Net_rb inference forward behind the C ABI.
//
// Host logic only.  demfi_ctx_bind lays every activation buffer out inside ONE caller-owned workspace, repacks the
// state_dict into MFMA fragment order (each layer once, shared by all contexts), builds one demfi_conv descriptor per
// convolution call site and records the launch sequence of the kernels of conv.hip / pointwise.hip.
//
// The plan follows the data flow of DeMFInet.forward (/root/reference/DeMFInet.py:46-179) but not its execution shape:
//   * every torch.cat is a multi-piece input of the consuming convolution (no concat buffers);
//   * RDB dense blocks grow in place, LFF outputs land directly in the 1152-channel GFF input;
//   * PixelShuffle / NN-upsample / tanh / sigmoid / ReLU / residual adds / GRU gate math are epilogues or fused loads;
//   * the t-independent trunk (FF_RDB + FAC-FB, 37 % of the MACs) is its own segment;
//   * Mixer.conv_ref1/2 do not depend on the recursion index and are hoisted out of the boosting loop.
// Flows, occlusion logits and 3-channel frames stay fp32 planar ("thin"); features are NHWC in the path dtype ("fat").
#include "common.h"
#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------------
// convolution descriptor builder (shared by demfi_conv_build and the plan)
// ---------------------------------------------------------------------------------------------------------------
constexpr int64_t LDS_BUDGET = 78 * 1024;      // general kernel: haloed tile + 2-tap weight ring, 2 workgroups per CU

struct BuiltConv {
    demfi_conv d;
    std::vector<uint8_t> wpack;
    std::vector<float> bias;
    int64_t wbytes = 0;        // size of the packed weights (also set when only sizing)
    int64_t macs = 0;
};

int build_conv(int dtype, int H, int W, int stride, int batch, const float* w, const float* bias, int cout, int cin, int kh,
               int kw, const demfi_conv_src* srcs, int n_srcs, const demfi_conv_dst* dsts, int n_dsts, BuiltConv& out,
               bool size_only, const char* name, int pad_y = -1, int pad_x = -1, int grid_div = 1)
{
    if (dtype != DEMFI_F16 && dtype != DEMFI_F32) return demfi_set_error(DEMFI_ERR_ARG, "%s: dtype", name);
    if (!srcs || !dsts || n_srcs <= 0 || n_dsts <= 0 || n_dsts > DEMFI_MAX_SEGS || (stride != 1 && stride != 2))
        return demfi_set_error(DEMFI_ERR_ARG, "%s: sources / destinations / stride", name);
    const int esz = dtype == DEMFI_F32 ? 4 : 2;
    const int64_t LH = 7 * stride + kh, LW = 31 * stride + kw;
    int n_oct = 0;
    for (int i = 0; i < n_dsts; ++i) n_oct += (dsts[i].n + 7) / 8;
    const int sub = (n_oct + 3) / 4;
    int nco = sub <= 5 ? sub : 4;
    int rec = 128;
    // round 6: the stride-2 4x4 layers (UNet encoders, DeMFInet.py:575-577) whose inputs are NHWC pieces of 32-channel multiples (+ at
    // most one 16-channel tail) and whose outputs are 64-channel blocks of one NHWC tensor belong to the phase-decomposed streamed-weight
    // kernel (wsconv.hip): units of 32 channels (64-byte records, a 16-channel tail padded to a whole unit), two 32-cout subtiles per
    // work item
    static const bool ws2 = !(getenv("DEMFI_WS2") && atoi(getenv("DEMFI_WS2")) == 0);
    // ... and so do the 3x3 stride-1 layers of that output shape with >= 96 input channels (the UNet decoders dec0 / dec1 / dec2 with their
    // upsampled pieces, FGAC's w_gen): the 64 -> 64 and the narrow layers keep their own kernels
    int src_nch = 0;
    for (int i = 0; i < n_srcs; ++i) src_nch += srcs[i].nch;
    const bool ws2_s2 = stride == 2 && kh == 4 && kw == 4, ws2_s1 = stride == 1 && kh == 3 && kw == 3 && src_nch > 64;
    bool ws2_shape = ws2 && esz == 2 && (ws2_s2 || ws2_s1) && pad_y < 0 && pad_x < 0 && n_dsts == 1 && dsts[0].n % 64 == 0 &&
                     dsts[0].mode == DEMFI_MODE_STORE && (dsts[0].act == DEMFI_ACT_NONE || dsts[0].act == DEMFI_ACT_RELU) && dsts[0].scale <= 1 &&
                     dsts[0].dst.sc == 1 && !dsts[0].dst.is_f32 && (!dsts[0].res.ptr || (dsts[0].res.sc == 1 && !dsts[0].res.is_f32));
    for (int i = 0; ws2_shape && i < n_srcs; ++i)
        // a tail unit at the end: a 16-channel piece, optionally followed by an 8-channel one (Dec_first_2: ref16 | agg3d)
        ws2_shape = srcs[i].fat && !srcs[i].v.is_f32 &&
                    (srcs[i].nch % 32 == 0 || (srcs[i].nch == 16 && !srcs[i].up_shift && (i == n_srcs - 1 || (i == n_srcs - 2 && srcs[n_srcs - 1].nch == 8))) ||
                     (srcs[i].nch == 8 && !srcs[i].up_shift && i == n_srcs - 1 && i > 0 && srcs[i - 1].nch == 16)) &&
                    (srcs[i].up_shift == 0 || (srcs[i].up_shift == 1 && ws2_s1 && H % 2 == 0 && W % 2 == 0));
    if (ws2_shape) nco = 2;
    // the SepConvGRU layers (1x5 / 5x1 over two 64-channel NHWC pieces) run on their own persistent kernel, which wants
    // the two pieces as two 64-channel chunks whatever the general kernel's LDS budget says
    bool sep = esz == 2 && stride == 1 && ((kh == 1 && kw == 5) || (kh == 5 && kw == 1)) && n_srcs == 2 && (cout == 64 || cout == 128);
    for (int i = 0; sep && i < n_srcs; ++i) sep = srcs[i].fat && srcs[i].nch == 64 && !srcs[i].up_shift;
    while (!sep && rec > 32 && LH * LW * (rec + 16) + 2 * (rec / 32) * nco * 1024 > LDS_BUDGET) rec /= 2;
    // Small grids (the half- and lower-resolution layers: 920 tiles at 720p): with 128-byte records only two workgroups fit a
    // CU (LDS), so ~1000 workgroups run in two rounds; 64-byte records (five per CU) finish in one.  Not for the single
    // 64-channel 3x3 shape, which belongs to the persistent kernel (it needs 128-byte records).
    {
        // grid_div: the batched per-t plan runs the layer over batch = images x contexts; the choice is made on the grid of ONE
        // context so that both plans use the same record size, i.e. the same summation order: bit-identical results
        const int64_t n_wg = (int64_t)((W + 31) / 32) * ((H + 7) / 8) * (batch / grid_div) * ((sub + nco - 1) / nco);
        const bool persist_shape = n_srcs == 1 && srcs[0].fat && srcs[0].nch == 64 && kh == 3 && kw == 3 && stride == 1;
        // measured at 720p (same box): the 48 RDB growth convs 0.045-0.072 -> 0.035-0.058 ms, dec2 0.082 -> 0.070; the 96-cout
        // layers (nco = 3: LFF, GFF.1) get slower with it, hence nco <= 2
        if (!sep && !persist_shape && rec == 128 && nco <= 2 && n_wg <= 5 * 256) rec = 64;
        // round 5: the RDB growth shape (3x3, <= 32 couts, >= 3 units of 32 channels from NHWC pieces) belongs to the 3x3 instantiation
        // of the streamed-weight kernel at any grid size: it walks 32-channel units (64-byte records)
        static const bool ws3 = !(getenv("DEMFI_WS3") && atoi(getenv("DEMFI_WS3")) == 0);
        bool rdb_shape = ws3 && !sep && esz == 2 && kh == 3 && kw == 3 && stride == 1 && sub == 1 && n_dsts == 1 && dsts[0].n == 32 && cin >= 96 && pad_y < 0 && pad_x < 0;
        for (int i = 0; rdb_shape && i < n_srcs; ++i) rdb_shape = srcs[i].fat && !srcs[i].up_shift && srcs[i].nch % 32 == 0;
        if (rdb_shape || ws2_shape) rec = 64;
    }

    // ---- every original input channel must be fed exactly once ------------------------------------------------
    {
        std::vector<int> seen(cin, 0);
        for (int i = 0; i < n_srcs; ++i)
            for (int j = 0; j < srcs[i].nch; ++j) {
                const int c = srcs[i].cin[j];
                if (c >= cin) return demfi_set_error(DEMFI_ERR_ARG, "%s: input map names channel %d >= cin %d", name, c, cin);
                if (c >= 0) seen[c]++;
            }
        for (int c = 0; c < cin; ++c)
            if (seen[c] != 1) return demfi_set_error(DEMFI_ERR_ARG, "%s: input channel %d fed %d times", name, c, seen[c]);
    }
    // ---- pack the input pieces into chunks of <= rec bytes (fat pieces first: 16-byte aligned) -------------------
    struct P { demfi_view v; int nch, lds_ch, up, fat; };
    struct Ck { int first, n, nks; };
    std::vector<P> pieces;
    std::vector<Ck> chunks;
    std::vector<int32_t> cin_map;
    int first = 0, fill = 0;
    const demfi_view null_view = {nullptr, 0, 0, 0, 0, 0, 0};
    auto close_chunk = [&]() {
        if (fill == 0) return;
        const int unit = ws2_shape ? 64 : 32;                     // wsconv.hip walks whole 32-channel units
        const int padb = (unit - fill % unit) % unit;
        if (padb) {
            pieces.push_back({null_view, padb / esz, fill / esz, 0, 0});
            cin_map.insert(cin_map.end(), padb / esz, -1);
            fill += padb;
        }
        chunks.push_back({first, (int)pieces.size() - first, fill / 32});
        first = (int)pieces.size();
        fill = 0;
    };
    std::vector<int> order;
    for (int i = 0; i < n_srcs; ++i) if (srcs[i].fat) order.push_back(i);
    for (int i = 0; i < n_srcs; ++i) if (!srcs[i].fat) order.push_back(i);
    for (int si : order) {
        const demfi_conv_src& s = srcs[si];
        const int selt = s.v.is_f32 ? 4 : 2;
        int done = 0;
        while (done < s.nch) {
            if (fill >= rec) close_chunk();
            const int room = (rec - fill) / esz;
            int take;
            if (s.fat) {
                if (fill % 16) {
                    const int padc = (16 - fill % 16) / esz;
                    pieces.push_back({null_view, padc, fill / esz, 0, 0});
                    cin_map.insert(cin_map.end(), padc, -1);
                    fill += padc * esz;
                    continue;
                }
                take = std::min(s.nch - done, room);
                int vec = take * esz / 16;
                if (vec == 0) { close_chunk(); continue; }
                int p2 = 1;
                while (p2 * 2 <= vec) p2 *= 2;                       // 1, 2, 4, 8 vectors per pixel
                take = p2 * 16 / esz;
            } else {
                take = std::min(s.nch - done, room);
            }
            demfi_view v = s.v;
            v.ptr = s.v.ptr ? (char*)s.v.ptr + (int64_t)done * s.v.sc * selt : nullptr;
            pieces.push_back({v, take, fill / esz, s.up_shift, s.fat ? 1 : 0});
            cin_map.insert(cin_map.end(), s.cin + done, s.cin + done + take);
            fill += take * esz;
            done += take;
        }
    }
    close_chunk();
    if ((int)chunks.size() > DEMFI_MAX_CHUNKS || (int)pieces.size() > DEMFI_MAX_PIECES)
        return demfi_set_error(DEMFI_ERR_ARG, "%s: %d chunks / %d pieces", name, (int)chunks.size(), (int)pieces.size());
    // ---- output routing ----------------------------------------------------------------------------------------
    struct Oct { int seg, n, ch; };
    std::vector<Oct> octs;
    std::vector<int32_t> cout_map;
    for (int si = 0; si < n_dsts; ++si) {
        const demfi_conv_dst& ds = dsts[si];
        for (int o = 0; o < ds.n; o += 8) {
            const int k = std::min(8, ds.n - o);
            octs.push_back({si, k, o});
            for (int j = 0; j < 8; ++j) cout_map.push_back(j < k ? ds.couts[o + j] : -1);
        }
    }
    {
        std::vector<int> seen(cout, 0);
        for (int c : cout_map) {
            if (c >= cout) return demfi_set_error(DEMFI_ERR_ARG, "%s: output map names channel %d >= cout %d", name, c, cout);
            if (c >= 0) seen[c]++;
        }
        for (int c = 0; c < cout; ++c)
            if (seen[c] != 1) return demfi_set_error(DEMFI_ERR_ARG, "%s: output channel %d routed %d times", name, c, seen[c]);
    }
    const int cout_pad = (sub + nco - 1) / nco * nco * 32;
    if (cout_pad > 256) return demfi_set_error(DEMFI_ERR_ARG, "%s: %d packed output channels > 256", name, cout_pad);
    while ((int)octs.size() < cout_pad / 8) {
        octs.push_back({0, 0, 0});
        cout_map.insert(cout_map.end(), 8, -1);
    }
    std::vector<int32_t> nks;
    for (auto& c : chunks) nks.push_back(c.nks);
    // ---- descriptor ----------------------------------------------------------------------------------------------
    demfi_conv& d = out.d;
    memset(&d, 0, sizeof(d));
    d.dtype = dtype; d.H = H; d.W = W;
    d.inH = stride == 2 ? H * stride : H;
    d.inW = stride == 2 ? W * stride : W;
    d.kh = kh; d.kw = kw; d.stride = stride;
    d.pad_y = pad_y >= 0 ? pad_y : (stride == 2 ? 1 : kh / 2);      // explicit: the 2x2 phase filters of an upsampled 3x3 layer
    d.pad_x = pad_x >= 0 ? pad_x : (stride == 2 ? 1 : kw / 2);
    d.batch = batch; d.cout_pad = cout_pad; d.nco = nco; d.rec_bytes = rec;
    d.n_chunks = (int)chunks.size(); d.n_pieces = (int)pieces.size(); d.n_segs = n_dsts;
    const int taps = kh * kw;
    int64_t tot_ks = 0;
    for (int k : nks) tot_ks += k;
    d.w_blk_stride = tot_ks * taps * nco * 64;
    int64_t woff = 0;
    for (size_t i = 0; i < chunks.size(); ++i) {
        d.chunks[i].first_piece = chunks[i].first;
        d.chunks[i].n_pieces = chunks[i].n;
        d.chunks[i].nks = chunks[i].nks;
        d.chunks[i].w_off = woff;
        woff += (int64_t)chunks[i].nks * taps * nco * 64;
    }
    for (size_t i = 0; i < pieces.size(); ++i) {
        d.pieces[i].v = pieces[i].v;
        d.pieces[i].nch = pieces[i].nch;
        d.pieces[i].lds_ch = pieces[i].lds_ch;
        d.pieces[i].up_shift = pieces[i].up;
        d.pieces[i].fat = pieces[i].fat;
    }
    for (int i = 0; i < n_dsts; ++i) {
        demfi_seg& sg = d.segs[i];
        sg.dst = dsts[i].dst; sg.res = dsts[i].res; sg.aux = dsts[i].aux;
        sg.act = dsts[i].act; sg.mode = dsts[i].mode;
        sg.scale = dsts[i].scale ? dsts[i].scale : 1;
        sg.dy = dsts[i].dy; sg.dx = dsts[i].dx;
    }
    for (size_t i = 0; i < octs.size(); ++i) { d.oct_seg[i] = octs[i].seg; d.oct_n[i] = octs[i].n; d.oct_ch[i] = octs[i].ch; }
    for (int sb = 0; sb < DEMFI_MAX_OCTS / 4; ++sb) d.sub_seg[sb] = -1;
    const bool f32 = dtype == DEMFI_F32;
    auto fat_ok = [&](const demfi_view& v) { return v.ptr && v.sc == 1 && (v.is_f32 != 0) == f32; };
    for (int sb = 0; sb < cout_pad / 32; ++sb) {
        const Oct* o4 = &octs[sb * 4];
        const int si = o4[0].seg;
        const demfi_conv_dst& ds = dsts[si];
        bool ok = o4[0].ch % 8 == 0;
        for (int j = 0; j < 4; ++j) ok = ok && o4[j].seg == si && o4[j].n == 8 && o4[j].ch == o4[0].ch + 8 * j;
        ok = ok && fat_ok(ds.dst) && (!ds.res.ptr || fat_ok(ds.res));
        if (ds.mode == DEMFI_MODE_GRU) ok = ok && fat_ok(ds.aux);
        if (ds.mode != DEMFI_MODE_STORE) ok = ok && ds.res.ptr;
        if (ok) d.sub_seg[sb] = si;
    }
    d.lw_magic = (uint32_t)((0x100000000ull + LW - 1) / LW);
    out.macs = (int64_t)cout * cin * taps * H * W * batch;
    // ---- weights / bias ----------------------------------------------------------------------------------------
    // Layers of the persistent kernels (conv.hip: 64-channel 3x3, narrow with an NHWC destination, SepConvGRU) are packed in their cout order: MFMA
    // row r of a 32-cout subtile holds channel (r>>4)*16 + ((r>>2)&1)*8 + ((r>>3)&1)*4 + (r&3), which makes the two accumulator
    // quads of a lane 8 consecutive channels (a 16-byte store without any cross-lane exchange).  The octet tables keep
    // describing the un-permuted routing (that kernel only reads oct_ch[0]).
    {
        demfi_conv probe = d;
        probe.zero_page = &probe;                                   // the context / caller sets the real one later
        if (!probe.pieces[0].v.ptr) probe.pieces[0].v.ptr = &probe; // sizing pass
        if (demfi_persist_eligible(&probe)) {
            d.cout_perm = 1;
            std::vector<int32_t> pm(cout_map.size());
            for (size_t i = 0; i < cout_map.size(); ++i) {
                const int sb = (int)i / 32, r = (int)i % 32;
                pm[i] = cout_map[sb * 32 + (r >> 4) * 16 + ((r >> 2) & 1) * 8 + ((r >> 3) & 1) * 4 + (r & 3)];
            }
            cout_map.swap(pm);
        }
    }
    int64_t nbytes = 0;
    int st = demfi_pack_conv_weights(w, cout, cin, kh, kw, cin_map.data(), (int)cin_map.size(), nks.data(), (int)nks.size(),
                                     cout_map.data(), cout_pad, nco, dtype, nullptr, &nbytes);
    if (st < 0) return st;
    out.wbytes = nbytes;
    out.wpack.resize(size_only ? 0 : nbytes);
    out.bias.assign(cout_pad, 0.0f);
    if (!size_only) {
        st = demfi_pack_conv_weights(w, cout, cin, kh, kw, cin_map.data(), (int)cin_map.size(), nks.data(), (int)nks.size(),
                                     cout_map.data(), cout_pad, nco, dtype, out.wpack.data(), &nbytes);
        if (st < 0) return st;
        for (int i = 0; i < cout_pad; ++i)
            if (cout_map[i] >= 0 && bias) out.bias[i] = bias[cout_map[i]];
    }
    return DEMFI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// the context
// ---------------------------------------------------------------------------------------------------------------
struct Tensor {
    int64_t off = -1;
    int kind = 0;              // 0 fat [B,h,w,C] path dtype, 1 thin [C,h,w] fp32, 2 raw int64
    int d[4] = {0, 0, 0, 0};   // fat: B,h,w,C; thin: C,h,w,1; raw: n,1,1,1
    int64_t bytes = 0;
    int64_t cstride = 0;       // per-t buffers: bytes between the copies of consecutive per-t contexts (tensor-major layout), 0 = trunk buffer
    int id = 0;                // > 0: index into demfi_ctx::id_cstride; views built from the tensor carry it in demfi_view._pad while the plan is built
};

// Liveness plan of one buffer set (round 5, the workspace arena): the big activation buffers whose first access in the launch
// sequence is a write share ONE arena; off = byte offset of the buffer's footprint inside the arena, size = arena bytes.
struct ArenaPlan { std::map<std::string, int64_t> off; int64_t size = 0; int64_t cstride = 0; bool on = false; };     // cstride: context stride of every arena member of a per-t set (the slot size)

struct Weight { std::vector<float> data; std::vector<int64_t> shape; };

struct Layer { int cout, cin, kh, kw; };

typedef std::map<std::string, Tensor> BufSet;
typedef std::vector<demfi_op> OpList;

}  // namespace

struct demfi_ctx {
    int H, W, N, dtype, n_trunk, n_ctx;
    int op_kind = 0, op_batch = 1;                     // 0: the DeMFI-Net forward; 1 / 2: a single-call operator context (SepConvGRU / FGAC, ABI v7)
    demfi_hparams hp;
    std::map<std::string, Weight> weights;
    std::map<std::string, Layer> table;
    // layout (computed at create time: offsets are relative to the workspace base)
    int64_t w_region = 0, w_bytes = 0, desc_off = 0, zero_off = 0, total = 0, n_descs = 0;
    std::vector<BufSet> tr_bufs;                       // [trunk]
    std::vector<std::vector<BufSet>> t_bufs;           // [trunk][c]
    std::vector<int64_t> id_cstride;                   // tensor id -> context stride in bytes (0: trunk buffer); [0] unused
    ArenaPlan arena_t, arena_tr;                       // liveness plans of the per-t / trunk buffer sets (empty: every buffer has its own memory)
    // after bind
    bool bound = false, on_host = false;
    char* base = nullptr;
    std::vector<demfi_conv> descs;
    std::vector<OpList> tr_ops;                        // [trunk]
    std::vector<std::vector<OpList>> head_ops;         // [trunk][c]
    std::vector<std::vector<std::vector<OpList>>> iter_ops;   // [trunk][c][it]
    // the same per-t segment as ONE launch sequence over all n_ctx contexts of a trunk set (demfi_forward_tb): every
    // convolution runs once with batch x n_ctx (the copies of a per-t buffer are contiguous), point-wise ops once per context
    std::vector<OpList> tb_head_ops;                   // [trunk]
    std::vector<std::vector<OpList>> tb_iter_ops;      // [trunk][it]
    // fusion decisions of the sizing pass ("<kind>:<name>" per fused launch, in plan order).  The arena gives the buffers a fused launch
    // never touches (the scratch between the convolutions of a residual block, the z buffer of a GRU half-step) NO memory, so the
    // bind pass must fuse exactly the same launches: checked in demfi_ctx_bind.
    std::vector<std::string> fused_dry, fused_now;
    std::vector<uint8_t> host_blob;                    // packed weights + biases staged on the host
    std::map<std::string, std::pair<int64_t, int64_t>> pack_cache;   // layer signature -> (w_off, b_off) inside the blob
    int64_t blob_fill = 0;
};

namespace {

int esz_of(const demfi_ctx* c) { return c->dtype == DEMFI_F32 ? 4 : 2; }

void layer_table(demfi_ctx* c)
{
    // the reference's registration order and shapes (DeMFInet.py:15-44, 189-231, 319-333, 361-378, 566-584, 770-868;
    // SURVEY.md Appendix A/B) -- mirrored by demfi_amd/spec.py for the module surface
    auto& t = c->table;
    if (c->op_kind == 1) {                                       // SepConvGRU (DeMFInet.py:830-836): keys of the reference module
        for (const char* g : {"z", "r", "q"}) t[std::string("conv") + g + "1"] = {64, 128, 1, 5};
        for (const char* g : {"z", "r", "q"}) t[std::string("conv") + g + "2"] = {64, 128, 5, 1};
        return;
    }
    if (c->op_kind == 2) {                                       // FGAC (DeMFInet.py:369-380); conv_source_k is accepted and dead at rr = 0
        t["conv_ref_k"] = {64, 64, 1, 1}; t["conv_source_k"] = {64, 64, 1, 1}; t["fusion"] = {64, 64, 1, 1};
        t["w_gen"] = {64, 128, 3, 3}; t["w_gen_2"] = {1, 64, 3, 3};
        return;
    }
    const int nf = c->hp.nf, r2 = c->hp.scale_factor * c->hp.scale_factor;
    const int G0 = 96, G = 32, Cn = 4, D = 12;
    auto add = [&](const std::string& n, int cout, int cin, int kh, int kw) { t[n] = {cout, cin, kh, kw}; };
    std::string p = "FF_RDB_Module.";
    add(p + "SFENet1", G0, 12 * r2, 5, 5);
    add(p + "SFENet2", G0, G0, 3, 3);
    for (int i = 0; i < D; ++i) {
        for (int k = 0; k < Cn; ++k) add(p + "RDBs." + std::to_string(i) + ".convs." + std::to_string(k) + ".conv.0", G, G0 + k * G, 3, 3);
        add(p + "RDBs." + std::to_string(i) + ".LFF", G0, G0 + Cn * G, 1, 1);
    }
    add(p + "GFF.0", G0, D * G0, 1, 1);
    add(p + "GFF.1", G0, G0, 3, 3);
    add(p + "UPNet.0", 256, G0, 3, 3);
    add(p + "UPNet.2", 2 * nf + 5, 64, 3, 3);
    p = "FAC_FB_Module.";
    add(p + "conv_first", nf, nf, 3, 3);
    for (int i = 0; i < c->hp.num_resb_facfb; ++i) {
        add(p + "feature_extraction." + std::to_string(i) + ".conv1", nf, nf, 3, 3);
        add(p + "feature_extraction." + std::to_string(i) + ".conv2", nf, nf, 3, 3);
    }
    std::vector<std::string> fg = c->hp.shared_fgac ? std::vector<std::string>{"shared_FGAC"}
                                                    : std::vector<std::string>{"FGAC_F1toF0", "FGAC_F0toF1"};
    for (auto& f : fg) {
        add(p + f + ".conv_ref_k", nf, nf, 1, 1);
        add(p + f + ".conv_source_k", nf, nf, 1, 1);
        add(p + f + ".w_gen", nf, 2 * nf, 3, 3);
        add(p + f + ".w_gen_2", 1, nf, 3, 3);
        add(p + f + ".fusion", nf, nf, 1, 1);
    }
    p = "Refine_Module.";
    add(p + "enc1", nf, 3 * nf + 9, 4, 4);
    add(p + "enc2", 2 * nf, nf, 4, 4);
    add(p + "enc3", 4 * nf, 2 * nf, 4, 4);
    add(p + "dec0", 4 * nf, 4 * nf, 3, 3);
    add(p + "dec1", 2 * nf, 6 * nf, 3, 3);
    add(p + "dec2", nf, 3 * nf, 3, 3);
    add(p + "dec3", 2 * nf + 5, nf, 3, 3);
    add("Dec_first", nf, nf, 3, 3);
    for (int i = 0; i < c->hp.num_resb_dec; ++i) {
        add("Decoder_res." + std::to_string(i) + ".conv1", nf, nf, 3, 3);
        add("Decoder_res." + std::to_string(i) + ".conv2", nf, nf, 3, 3);
    }
    add("Dec_last1", nf, nf, 3, 3);
    add("Dec_last2", 3, nf, 3, 3);
    add("Ch_Reducer", nf, 3 * nf, 7, 7);
    p = "Booster_Module.";
    add(p + "Mixer.conv_ref1", nf / 2, 30, 7, 7);
    add(p + "Mixer.conv_ref2", nf / 2, nf / 2, 3, 3);
    add(p + "Mixer.conv_delta1", nf / 2, 5, 7, 7);
    add(p + "Mixer.conv_delta2", nf / 2, nf / 2, 3, 3);
    add(p + "Mixer.conv_blend1", nf / 2, nf, 3, 3);
    add(p + "Mixer.conv_blend2", nf, nf / 2, 3, 3);
    for (const char* g : {"z", "r", "q"}) add(p + "GB.conv" + g + "1", nf, 2 * nf, 1, 5);
    for (const char* g : {"z", "r", "q"}) add(p + "GB.conv" + g + "2", nf, 2 * nf, 5, 1);
    add(p + "flow_occ.conv1", nf / 2, nf, 3, 3);
    add(p + "flow_occ.conv2", 5, nf / 2, 3, 3);
    add("Dec_first_2", nf, 9 + nf + 9 + 5 + 12, 3, 3);
    for (int i = 0; i < c->hp.num_resb_dec; ++i) {
        add("Decoder_res_2." + std::to_string(i) + ".conv1", nf, nf, 3, 3);
        add("Decoder_res_2." + std::to_string(i) + ".conv2", nf, nf, 3, 3);
    }
    add("Dec_last1_2", nf, nf, 3, 3);
    add("Dec_last2_2", 9, nf, 3, 3);
}

// ---- buffer layout ---------------------------------------------------------------------------------------------
struct Layout {
    demfi_ctx* c;
    int64_t cur;
    int rep = 0;               // > 0: per-t buffers, `rep` copies of every buffer back to back (copy q at off + q * cstride)
    const ArenaPlan* plan = nullptr;   // buffers named in it live at arena_base + their planned offset
    int64_t arena_base = 0;
    int64_t take(int64_t bytes) { const int64_t o = cur; cur = (cur + bytes + 255) & ~255ll; return o; }
    static int64_t footprint(const Tensor& t, int rep) { return rep > 0 ? ((t.bytes + 15) & ~15ll) * rep : t.bytes; }
    void place(Tensor& t, const char* n)
    {
        if (rep > 0) t.cstride = (t.bytes + 15) & ~15ll;
        const int64_t fp = footprint(t, rep);
        auto it = plan && plan->on ? plan->off.find(n) : std::map<std::string, int64_t>::const_iterator();
        if (plan && plan->on && it != plan->off.end()) {
            t.off = arena_base + it->second;
            if (rep > 0) t.cstride = plan->cstride;              // arena members of a per-t set: copy q sits q SLOTS further (see plan_arena)
        } else t.off = take(fp);
        t.id = (int)c->id_cstride.size();
        c->id_cstride.push_back(t.cstride);
    }
    void begin_set(const ArenaPlan* pl)
    {
        plan = pl;
        if (pl && pl->on) arena_base = take(pl->size);
    }
    void fat(BufSet& s, const char* n, int h, int w, int ch, int b = 1)
    {
        Tensor t; t.kind = 0; t.d[0] = b; t.d[1] = h; t.d[2] = w; t.d[3] = ch;
        t.bytes = (int64_t)b * h * w * ch * esz_of(c); place(t, n); s[n] = t;
    }
    void thin(BufSet& s, const char* n, int ch, int h, int w)
    {
        Tensor t; t.kind = 1; t.d[0] = ch; t.d[1] = h; t.d[2] = w; t.d[3] = 1;
        t.bytes = (int64_t)ch * h * w * 4; place(t, n); s[n] = t;
    }
    void raw(BufSet& s, const char* n, int64_t bytes)
    {
        Tensor t; t.kind = 2; t.d[0] = (int)(bytes / 8); t.d[1] = t.d[2] = t.d[3] = 1;
        t.bytes = bytes; place(t, n); s[n] = t;
    }
};

void alloc_trunk(Layout& L, BufSet& s)
{
    const int H = L.c->H, W = L.c->W, H2 = H / 2, W2 = W / 2;
    L.thin(s, "x", 12, H, W);                       // module input [3,4,H,W], batch 1
    L.fat(s, "s2d", H2, W2, 48);
    L.fat(s, "f1", H2, W2, 96);
    L.fat(s, "x0", H2, W2, 96);
    L.fat(s, "grow", H2, W2, 128);
    L.fat(s, "gffcat", H2, W2, 1152);
    L.fat(s, "g0", H2, W2, 96);
    L.fat(s, "g1", H2, W2, 96);
    L.fat(s, "up", H, W, 64);
    L.fat(s, "F01", H, W, 64, 2);
    L.thin(s, "ffo", 5, H, W);                      // flow_01 (2), flow_10 (2), occ_0 logit (1)
    L.fat(s, "enc_a", H, W, 64, 2);
    L.fat(s, "enc_t", H, W, 64, 2);
    L.fat(s, "enc_b", H, W, 64, 2);
    L.fat(s, "rk", H, W, 64, 2);
    if (L.c->hp.fgac_rr > 0) {
        L.fat(s, "skk", H, W, 64, 2);               // conv_source_k(source): live only in the generalised FGAC
        if (L.c->hp.fgac_sr > 0) { L.fat(s, "rkp", H, W, 64, 2); L.fat(s, "skp", H, W, 64, 2); }
    }
    L.fat(s, "smp", H, W, 64, 2);
    L.fat(s, "E", H, W, 64, 2);
    L.fat(s, "wg", H, W, 64, 2);
    L.thin(s, "gate", 2, H, W);
    L.fat(s, "aF", H, W, 64, 2);
    L.thin(s, "overlay", 3, H, W);
    if (L.c->hp.flags & DEMFI_HP_EXTRAS) {
        // per FGAC direction b (0: F1 -> F0, 1: F0 -> F1), planes 6 b + {0: 1 - w_sr, 1: source_v, 2: init_ref_k, 3: E_s, 4: bolstered_F_s,
        // 5: diff}: the min-max normalised channel means of DeMFInet.py:454-494 (w_sr itself is the "gate" buffer)
        L.thin(s, "viz", 12, H, W);
        L.thin(s, "vizs", 1, 1, (int)demfi_minmax_scratch_floats());
    }
    if (L.c->dtype == DEMFI_F16) {
        L.fat(s, "u1a", H2, W2, 64);                // t-independent part of Refine_Module.enc1 (see build_trunk)
        L.fat(s, "xff16", H, W, 16);                // window-constant planes of the Mixer / D2 inputs: 4 frames x 3 colours | flow_10, flow_01
        L.fat(s, "re1w", H, W, 32);                 // their share of Mixer.conv_ref1 ...
        L.fat(s, "g_pw", H, W, 64);                 // ... and of Dec_first_2
    }
}

// buffers of a single-call operator context (all in the "trunk" set: demfi_ctx_buffer(ctx, 0, -1, name, ...))
void alloc_operator(Layout& L, BufSet& s)
{
    const int H = L.c->H, W = L.c->W, B = L.c->op_batch;
    if (L.c->op_kind == 1) {
        for (const char* n : {"h", "x", "z", "rh", "h1", "out"}) L.fat(s, n, H, W, 64, B);
    } else {
        for (const char* n : {"ref", "source", "ref_k", "sampled", "e_s", "hid", "out"}) L.fat(s, n, H, W, 64, B);
        L.thin(s, "flow", 2 * B, H, W);                          // flow_s2r [B,2,H,W] fp32 (absolute sampling coordinates)
        L.thin(s, "w", B, H, W);                                 // the gate w_sr [B,1,H,W]
    }
}

void alloc_t(Layout& L, BufSet& s)
{
    const int H = L.c->H, W = L.c->W, N = L.c->N;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    L.thin(s, "t", 1, 1, 1);
    L.raw(s, "sink", 256);                          // demfi_u8_sink record (zero = disabled: iter 0 with NULL frames writes nothing)
    L.raw(s, "cfr_acc", demfi_cfr_workspace_bytes(H, W));
    L.thin(s, "ft", 4, H, W);                       // flow_t0, flow_t1
    L.fat(s, "Ft", H, W, 64);
    L.fat(s, "u1", H2, W2, 64);
    L.fat(s, "u2", H4, W4, 128);
    L.fat(s, "u3", H8, W8, 256);
    L.fat(s, "d0", H8, W8, 256);
    L.fat(s, "d1", H4, W4, 128);
    L.fat(s, "d2", H2, W2, 64);
    L.fat(s, "rF", H, W, 64, 3);                    // rF0, rF1, rFt
    L.thin(s, "delta", 5 * (N + 1), H, W);          // (flow_t0, flow_t1, occ logit) per step
    L.thin(s, "occ", N + 1, H, W);                  // sigmoid(occ logit) per step
    L.fat(s, "dec_a", H, W, 64, 3);
    L.fat(s, "dec_t", H, W, 64, 3);
    L.fat(s, "dec_b", H, W, 64, 3);
    L.thin(s, "sharp1", 9, H, W);                   // S0p, S1p, Stp
    L.fat(s, "frec0", H, W, 64);
    L.fat(s, "frec1", H, W, 64);
    L.fat(s, "re1", H, W, 32);
    // Mixer.conv_ref2 | conv_delta2 outputs as the two 32-channel halves of ONE 64-channel buffer: conv_blend1 (cat[ref, delta],
    // DeMFInet.py:826-827) then stages a single 128-byte record per pixel (the narrow kernel's fast DMA path) instead of two pieces
    L.fat(s, "rd64", H, W, 64);
    L.fat(s, "de1", H, W, 32);
    L.fat(s, "bl1", H, W, 32);
    L.fat(s, "xb", H, W, 64);
    L.fat(s, "zb", H, W, 64);
    L.fat(s, "rh", H, W, 64);
    L.fat(s, "h1", H, W, 64);
    L.fat(s, "fo1", H, W, 32);
    L.thin(s, "stnew", 3, H, W);
    // planar flows / logits / frames packed to NHWC once, so the consuming convs stage them with vector loads
    L.fat(s, "misc16", H, W, 16);
    if (L.c->dtype == DEMFI_F16) L.fat(s, "ref16", H, W, 16);     // per-t planes only (fp16 plan): S0p, S1p, Stp | rflow_t0, rflow_t1, occ logit | occ_0 | 0
    else { L.fat(s, "ref32", H, W, 32); L.fat(s, "agg3s", H, W, 32); }
    L.fat(s, "agg3d", H, W, 8);
    L.fat(s, "delta16", H, W, 16);               // 5 flow / occlusion planes + 11 zero channels: a full 32-byte record (one DMA piece)
    L.fat(s, "g_a", H, W, 64);
    L.fat(s, "g_t", H, W, 64);
    L.fat(s, "g_b", H, W, 64);
    if (L.c->dtype == DEMFI_F16) L.fat(s, "g_p2", H, W, 64);     // partial sum of Dec_first_2 (everything but the F_rec part)
    L.thin(s, "finals", 9 * N, H, W);               // [N][3 frames][3 colours]
}

void compute_layout(demfi_ctx* c, int64_t w_bytes, int64_t n_descs)
{
    Layout L{c, 0};
    c->w_region = L.take(0);
    c->w_bytes = (w_bytes + 255) & ~255ll;
    L.take(c->w_bytes);
    c->zero_off = L.take(256);
    c->n_descs = n_descs;
    c->desc_off = L.take(n_descs * (int64_t)sizeof(demfi_conv));
    c->tr_bufs.assign(c->n_trunk, BufSet());
    c->t_bufs.assign(c->n_trunk, std::vector<BufSet>(c->n_ctx));
    c->id_cstride.assign(1, 0);
    if (c->op_kind) {
        alloc_operator(L, c->tr_bufs[0]);
        c->total = L.cur;
        return;
    }
    for (int k = 0; k < c->n_trunk; ++k) {
        L.begin_set(&c->arena_tr);
        alloc_trunk(L, c->tr_bufs[k]);
        // tensor-major: the n_ctx copies of a per-t buffer are contiguous, so a convolution over "batch x n_ctx" addresses
        // all of them with one batch stride (demfi_forward_tb)
        L.rep = c->n_ctx;
        L.begin_set(&c->arena_t);
        alloc_t(L, c->t_bufs[k][0]);
        L.rep = 0;
        L.begin_set(nullptr);
        for (int q = 1; q < c->n_ctx; ++q) {
            c->t_bufs[k][q] = c->t_bufs[k][0];
            for (auto& kv : c->t_bufs[k][q]) kv.second.off += q * kv.second.cstride;
        }
    }
    c->total = L.cur;
}

// ---- plan builder ----------------------------------------------------------------------------------------------
struct Src { demfi_view v; int fat, up; std::vector<int32_t> cin; };
struct Dst { demfi_view dst, res, aux; int act, mode, scale, dy, dx; std::vector<int32_t> couts; };

std::vector<int32_t> range(int a, int b) { std::vector<int32_t> r; for (int i = a; i < b; ++i) r.push_back(i); return r; }
const demfi_view NOVIEW = {nullptr, 0, 0, 0, 0, 0, 0};

struct Builder {
    demfi_ctx* c;
    int esz;
    bool f32;
    bool dry;                  // sizing pass of demfi_ctx_create: no weights, nothing is written
    int status = DEMFI_OK;
    const demfi_u8_sink* sink_for_next = nullptr;   // uint8 sink record of the NEXT conv() call (Dec_last2_2)
    int sink_iter = 0;
    // packed copy of the NEXT conv() call's thin outputs (demfi_conv.pack): NHWC view + channel of each octet (-1 = not packed)
    demfi_view pack_for_next = {nullptr, 0, 0, 0, 0, 0, 0};
    int pack_ch_for_next[4] = {-1, -1, -1, -1};
    void pack_next(demfi_view v, int c0, int c1 = -1, int c2 = -1, int c3 = -1)
    {
        pack_for_next = v;
        pack_ch_for_next[0] = c0; pack_ch_for_next[1] = c1; pack_ch_for_next[2] = c2; pack_ch_for_next[3] = c3;
    }
    // ---- batched per-t plan (demfi_forward_tb): build_t on context 0 of trunk set tb_k with tb = n_ctx -----------------
    int tb = 1, tb_k = 0;
    // the buffer a device pointer lies in: per-t buffer of context 0 (returns its context stride in bytes), trunk buffer (0),
    // or neither (-1: weights, zero page, NULL)
    int64_t ctx_stride_of(const void* p) const
    {
        if (!p) return -1;
        const int64_t off = (const char*)p - c->base;
        for (const auto& kv : c->t_bufs[tb_k][0])
            if (off >= kv.second.off && off < kv.second.off + kv.second.bytes) return kv.second.cstride;
        for (const auto& kv : c->tr_bufs[tb_k])
            if (off >= kv.second.off && off < kv.second.off + kv.second.bytes) return 0;
        return -1;
    }
    // context stride of a view: by the id of the tensor it was built from (buffers of the arena share addresses, so an address does
    // not name a buffer any more); raw pointers (thin planes: never in the arena) by address
    int64_t view_stride(const demfi_view& v) const
    {
        if (v._pad > 0 && v._pad < (int)c->id_cstride.size()) return c->id_cstride[v._pad];
        return ctx_stride_of(v.ptr);
    }
    // view of a convolution of the batched plan: the conv runs with batch nb * tb, image index = q * nb + f
    bool tb_view(demfi_view& v, int nb, const char* name)
    {
        if (!v.ptr) return true;
        const int64_t cs = view_stride(v), elt = v.is_f32 ? 4 : 2;
        v._pad = 0;
        if (cs < 0) { status = demfi_set_error(DEMFI_ERR_ARG, "%s: view outside the context's buffers in the batched plan", name); return false; }
        if (cs == 0) {                                           // trunk buffer: the same image for every context
            if (nb != 1 && v.sb != 0) { status = demfi_set_error(DEMFI_ERR_ARG, "%s: batched trunk view in a batch-%d layer", name, nb); return false; }
            v.sb = 0;
        } else if (nb == 1) v.sb = cs / elt;                     // one image per context
        else if (v.sb * nb * elt != cs) {                        // nb images per context: they must tile the context stride
            status = demfi_set_error(DEMFI_ERR_ARG, "%s: %d images of stride %lld do not tile the context stride %lld", name, nb,
                                     (long long)(v.sb * elt), (long long)cs);
            return false;
        }
        return true;
    }
    const void* tb_ptr(const void* p, int q) const
    {
        const int64_t cs = ctx_stride_of(p);
        return cs > 0 ? (const char*)p + q * cs : p;
    }

    char* ptr(const Tensor& t) const { return c->base + t.off; }
    // input piece from a fat buffer [B,h,w,C]: channels [c0, c0+nch) feed original cin [cin0, cin0+nch); b < 0 keeps the
    // batch stride (batched conv), b >= 0 pins image b
    Src fsrc(const Tensor& t, int cin0, int c0 = 0, int nch = -1, int b = -1, int up = 0) const
    {
        const int h = t.d[1], w = t.d[2], Ct = t.d[3];
        if (nch < 0) nch = Ct - c0;
        Src s;
        s.v = {ptr(t) + ((int64_t)c0 + (b < 0 ? 0 : (int64_t)b * h * w * Ct)) * esz, Ct, (int64_t)w * Ct, 1,
               b < 0 ? (int64_t)h * w * Ct : 0, f32 ? 1 : 0, t.id};
        s.fat = 1; s.up = up; s.cin = range(cin0, cin0 + nch);
        return s;
    }
    // ALL channels of a fat buffer with an explicit channel -> original-cin list (-1 = unused padding channel)
    Src fsrc_map(const Tensor& t, const std::vector<int32_t>& cin, int b = 0) const
    {
        const int h = t.d[1], w = t.d[2], Ct = t.d[3];
        Src s;
        s.v = {ptr(t) + (int64_t)b * h * w * Ct * esz, Ct, (int64_t)w * Ct, 1, 0, f32 ? 1 : 0, t.id};
        s.fat = 1; s.up = 0; s.cin = cin;
        return s;
    }
    demfi_view fview(const Tensor& t, int c0 = 0, int b = -1) const
    {
        const int h = t.d[1], w = t.d[2], Ct = t.d[3];
        return {ptr(t) + ((int64_t)c0 + (b < 0 ? 0 : (int64_t)b * h * w * Ct)) * esz, Ct, (int64_t)w * Ct, 1,
                b < 0 ? (int64_t)h * w * Ct : 0, f32 ? 1 : 0, t.id};
    }
    demfi_view tview(const Tensor& t, int c0 = 0, int64_t sb = 0) const
    {
        const int h = t.d[1], w = t.d[2];
        return {ptr(t) + (int64_t)c0 * h * w * 4, 1, w, (int64_t)h * w, sb, 1, t.id};
    }
    const float* plane(const Tensor& t, int ch) const { return (const float*)(ptr(t) + (int64_t)ch * t.d[1] * t.d[2] * 4); }
    static Dst D(demfi_view v, std::vector<int32_t> couts, int act = DEMFI_ACT_NONE, int mode = DEMFI_MODE_STORE,
                 demfi_view res = NOVIEW, demfi_view aux = NOVIEW, int scale = 1, int dy = 0, int dx = 0)
    {
        return Dst{v, res, aux, act, mode, scale, dy, dx, std::move(couts)};
    }

    int64_t blob_put(const void* p, int64_t n)
    {
        const int64_t off = c->blob_fill;
        if (!dry) {
            if (off + n > c->w_bytes) { status = demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: weight region too small"); return 0; }
            memcpy(c->host_blob.data() + off, p, n);
        }
        c->blob_fill = (off + n + 255) & ~255ll;
        return off;
    }

    void conv(OpList& seg, const std::string& name, const std::vector<Src>& srcs, const std::vector<Dst>& dsts, int H, int W,
              int stride = 1, int batch = 1, const std::vector<float>* wt = nullptr, const std::vector<float>* bs = nullptr,
              const Layer* shape = nullptr, int pad_y = -1, int pad_x = -1)
    {
        if (status < 0) return;
        Layer l;
        const float *w, *b;
        static const float dummy = 0.0f;
        if (dry) {
            if (shape) l = *shape;
            else {
                auto it = c->table.find(name);
                if (it == c->table.end()) { status = demfi_set_error(DEMFI_ERR_ARG, "unknown layer '%s'", name.c_str()); return; }
                l = it->second;
            }
            w = b = &dummy;
        } else if (wt) { w = wt->data(); b = bs->data(); l = *shape; }
        else {
            auto it = c->table.find(name);
            auto iw = c->weights.find(name + ".weight"), ib = c->weights.find(name + ".bias");
            if (it == c->table.end() || iw == c->weights.end() || ib == c->weights.end()) {
                status = demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: weight '%s' was not loaded", name.c_str());
                return;
            }
            l = it->second; w = iw->second.data.data(); b = ib->second.data.data();
        }
        std::vector<demfi_conv_src> cs(srcs.size());
        for (size_t i = 0; i < srcs.size(); ++i) cs[i] = {srcs[i].v, srcs[i].fat, srcs[i].up, (int32_t)srcs[i].cin.size(), 0, srcs[i].cin.data()};
        std::vector<demfi_conv_dst> cd(dsts.size());
        for (size_t i = 0; i < dsts.size(); ++i)
            cd[i] = {dsts[i].dst, dsts[i].res, dsts[i].aux, dsts[i].act, dsts[i].mode, dsts[i].scale, dsts[i].dy, dsts[i].dx,
                     (int32_t)dsts[i].couts.size(), dsts[i].couts.data()};
        demfi_view pack_v = pack_for_next;
        pack_for_next.ptr = nullptr;
        if (tb > 1) {
            if (!tb_view(pack_v, batch, name.c_str())) return;
            for (auto& x : cs) if (!tb_view(x.v, batch, name.c_str())) return;
            for (auto& x : cd) if (!tb_view(x.dst, batch, name.c_str()) || !tb_view(x.res, batch, name.c_str()) || !tb_view(x.aux, batch, name.c_str())) return;
            batch *= tb;
        }
        pack_v._pad = 0;                                         // the tensor ids are the builder's business, not the descriptors'
        for (auto& x : cs) x.v._pad = 0;
        for (auto& x : cd) x.dst._pad = x.res._pad = x.aux._pad = 0;
        // the packed blob of a call site depends on its channel maps only (not on buffer addresses): per-t contexts and
        // the two FGAC directions share one copy
        std::string sig = name + "|";
        for (auto& s : srcs) { sig += s.fat ? 'F' : 'T'; for (int32_t ch : s.cin) sig += std::to_string(ch) + ","; sig += ';'; }
        sig += "|";
        for (auto& d : dsts) { for (int32_t ch : d.couts) sig += std::to_string(ch) + ","; sig += ';'; }
        BuiltConv bc;
        status = build_conv(c->dtype, H, W, stride, batch, w, b, l.cout, l.cin, l.kh, l.kw, cs.data(), (int)cs.size(), cd.data(),
                            (int)cd.size(), bc, true, name.c_str(), pad_y, pad_x, tb);    // descriptor + sizes
        if (status < 0) return;
        // which kernel owns the layer decides the packed cout order; the record size / cout blocking (they depend on the grid,
        // i.e. on the batch: the batched plan may choose differently) decide the chunk order of the blob
        sig += bc.d.cout_perm ? "|P" : "|N";
        sig += "|r" + std::to_string(bc.d.rec_bytes) + "n" + std::to_string(bc.d.nco);
        auto hit = c->pack_cache.find(sig);
        if (!dry && hit == c->pack_cache.end()) {
            bc = BuiltConv();
            status = build_conv(c->dtype, H, W, stride, batch, w, b, l.cout, l.cin, l.kh, l.kw, cs.data(), (int)cs.size(), cd.data(),
                                (int)cd.size(), bc, false, name.c_str(), pad_y, pad_x, tb);
            if (status < 0) return;
        }
        int64_t w_off, b_off;
        if (hit != c->pack_cache.end()) { w_off = hit->second.first; b_off = hit->second.second; }
        else {
            w_off = blob_put(bc.wpack.data(), bc.wbytes);
            b_off = blob_put(bc.bias.data(), (int64_t)bc.bias.size() * 4);
            if (status < 0) return;
            c->pack_cache[sig] = {w_off, b_off};
        }
        bc.d.wpack = c->base + c->w_region + w_off;
        bc.d.bias = (const float*)(c->base + c->w_region + b_off);
        bc.d.zero_page = c->base + c->zero_off;
        bc.d.u8_sink = sink_for_next;                // set by the caller for the frame-producing layer only
        bc.d.u8_iter = sink_iter;
        sink_for_next = nullptr;
        bc.d.pack = pack_v;
        for (int g = 0; g < 4; ++g) bc.d.pack_oct_ch[g] = pack_v.ptr ? pack_ch_for_next[g] : -1;
        c->descs.push_back(bc.d);
        demfi_op op;
        memset(&op, 0, sizeof(op));
        op.kind = DEMFI_OP_CONV;
        op.conv = (int)c->descs.size() - 1;
        op.macs = bc.macs;
        strncpy(op.name, name.c_str(), sizeof(op.name) - 1);
        seg.push_back(op);
    }

    // Sub-convolution over the original input channels `sel` (in that order) of layer `name`: a convolution is linear in its
    // input channels, so conv(cat[A, B]) = conv_A(A) + conv_B(B); the fp16 plan uses it to hoist the part of a layer whose
    // inputs do not change (per window / per recursion) and to bring the rest onto the persistent kernels.
    struct SubW { std::vector<float> w, b; Layer shape; };
    // rows [co0, co0 + n) of a layer's weight / bias: one launch per group of output channels (UPNet.2)
    SubW sub_weight_cout(const std::string& name, int co0, int n)
    {
        SubW o;
        auto it = c->table.find(name);
        if (it == c->table.end()) { status = demfi_set_error(DEMFI_ERR_ARG, "unknown layer '%s'", name.c_str()); return o; }
        const Layer& l = it->second;
        o.shape = {n, l.cin, l.kh, l.kw};
        if (dry) return o;
        auto iw = c->weights.find(name + ".weight"), ib = c->weights.find(name + ".bias");
        if (iw == c->weights.end() || ib == c->weights.end()) {
            status = demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: weight '%s' was not loaded", name.c_str());
            return o;
        }
        const size_t row = (size_t)l.cin * l.kh * l.kw;
        o.w.assign(iw->second.data.begin() + co0 * row, iw->second.data.begin() + (co0 + n) * row);
        o.b.assign(ib->second.data.begin() + co0, ib->second.data.begin() + co0 + n);
        return o;
    }
    SubW sub_weight(const std::string& name, const std::vector<int32_t>& sel, bool with_bias)
    {
        SubW o;
        auto it = c->table.find(name);
        if (it == c->table.end()) { status = demfi_set_error(DEMFI_ERR_ARG, "unknown layer '%s'", name.c_str()); return o; }
        const Layer& l = it->second;
        o.shape = {l.cout, (int)sel.size(), l.kh, l.kw};
        if (dry) return o;
        auto iw = c->weights.find(name + ".weight"), ib = c->weights.find(name + ".bias");
        if (iw == c->weights.end() || ib == c->weights.end()) {
            status = demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: weight '%s' was not loaded", name.c_str());
            return o;
        }
        const int taps = l.kh * l.kw;
        o.w.resize((size_t)l.cout * sel.size() * taps);
        for (int co = 0; co < l.cout; ++co)
            for (size_t k = 0; k < sel.size(); ++k)
                memcpy(&o.w[((size_t)co * sel.size() + k) * taps], &iw->second.data[((size_t)co * l.cin + sel[k]) * taps], taps * sizeof(float));
        o.b.assign(l.cout, 0.0f);
        if (with_bias) o.b = ib->second.data;
        return o;
    }

    void simple(OpList& seg, int kind, const char* name, demfi_op op)
    {
        op.kind = kind;
        strncpy(op.name, name, sizeof(op.name) - 1);
        const int64_t cs_a = op.a.ptr ? view_stride(op.a) : -1, cs_b = op.b.ptr ? view_stride(op.b) : -1, cs_o = op.o.ptr ? view_stride(op.o) : -1;
        op.a._pad = op.b._pad = op.o._pad = 0;
        if (tb <= 1) { seg.push_back(op); return; }
        // batched plan.  CFR and the thin (3-channel) warps: ONE launch for all tb per-t contexts (ABI v5, demfi_batch): the pointers
        // are those of context 0, every pointer gets the byte stride of the buffer it lies in (per-t buffers: their context stride;
        // window-level buffers of the trunk set: 0).  Measured in sequence at 720p x 7 contexts (profiles/r03_notes.md): cfr 81 -> 67 us
        // and warp_thin 37 -> 35 us per time instant.  The fat warp and the plane packs stay one launch per context: batched they
        // were SLOWER (pack 23 -> 31 us per context; fat warp with the contexts innermost per tile 85 -> 73 / 90 us: the gathered
        // neighbourhoods of a tile do not survive in the 4 MB L2 across seven time instants with these incoherent flows).
        // Round 5: the fat warps too, as ONE launch with one grid slice per context (demfi_batch._pad = 1): the same tiles in the same
        // order as tb launches, without their launch gaps and tails (a launch is ~80 us; DEMFI_WARP_TB=0: one launch per context,
        // 2: contexts innermost per tile, the round-3 form that was slower for the rF warps)
        static const int warp_tb = getenv("DEMFI_WARP_TB") ? atoi(getenv("DEMFI_WARP_TB")) : 1;
        const bool fat_warp = kind == DEMFI_OP_WARP && op.nch != 3 && warp_tb != 0;
        static const int pack_tb = getenv("DEMFI_PACK_TB") ? atoi(getenv("DEMFI_PACK_TB")) : 1;   // plane packs as one launch, grid.y = context (22 -> 4 launches per window: -0.1 ms; 0 = one launch per context)
        const bool one_launch = kind == DEMFI_OP_CFR || (kind == DEMFI_OP_WARP && op.nch == 3) || fat_warp || (kind == DEMFI_OP_PACK && pack_tb);
        if (one_launch) {
            op.bt._pad = fat_warp && warp_tb == 1 ? 1 : 0;
            auto stride = [&](const void* p) { const int64_t cs = ctx_stride_of(p); return cs > 0 ? cs : (int64_t)0; };
            op.bt.nb = tb;
            op.bt.a = cs_a > 0 ? cs_a : 0; op.bt.b = cs_b > 0 ? cs_b : 0; op.bt.o = cs_o > 0 ? cs_o : 0; op.bt.t = stride(op.t);
            for (int i = 0; i < 32; ++i) op.bt.p[i] = stride(op.p[i]);
            seg.push_back(op);
            return;
        }
        for (int q = 0; q < tb; ++q) {                          // one launch per context, pointers rebased
            demfi_op o = op;
            if (op.a.ptr && cs_a > 0) o.a.ptr = (char*)op.a.ptr + q * cs_a;
            if (op.b.ptr && cs_b > 0) o.b.ptr = (char*)op.b.ptr + q * cs_b;
            if (op.o.ptr && cs_o > 0) o.o.ptr = (char*)op.o.ptr + q * cs_o;
            for (int i = 0; i < 32; ++i) o.p[i] = tb_ptr(op.p[i], q);
            o.t = tb_ptr(op.t, q);
            seg.push_back(o);
        }
    }
    static demfi_op blank() { demfi_op o; memset(&o, 0, sizeof(o)); return o; }

    void pack(OpList& seg, const std::vector<const float*>& planes, const Tensor& dst)
    {
        demfi_op op = blank();
        op.nch = dst.d[3];
        for (int i = 0; i < 32; ++i) op.p[i] = i < (int)planes.size() ? planes[i] : nullptr;
        op.o = fview(dst);
        simple(seg, DEMFI_OP_PACK, "pack", op);
    }

    // Round 5: the two launches conv() has just appended (conv1 -> ReLU -> t, conv2 + identity) become ONE launch of the fused
    // residual-block kernel when the pair qualifies (fp16 plan, 3x3 64 -> 64, persistent-kernel packing): the intermediate stays in
    // LDS, the scratch buffer t is not touched -- and under the workspace arena it has NO memory (its views point at the arena's first
    // bytes, which belong to a live tenant): a RESBLOCK op must never be executed as its two convolutions on the bound workspace.  Both
    // descriptors are kept as they are for the CPU plan interpreter (which gives the intermediate private memory, tests/plan_sim.py)
    // and for DEMFI_RESBLOCK=0, which changes the sizing pass too (the scratch then has memory).
    void fuse_resblock(OpList& seg, const std::string& name)
    {
        static const bool on = !(getenv("DEMFI_RESBLOCK") && atoi(getenv("DEMFI_RESBLOCK")) == 0);
        if (status < 0 || !on || seg.size() < 2) return;
        const demfi_op o2 = seg[seg.size() - 1], o1 = seg[seg.size() - 2];
        if (o1.kind != DEMFI_OP_CONV || o2.kind != DEMFI_OP_CONV) return;
        demfi_conv h1 = c->descs[o1.conv], h2 = c->descs[o2.conv];
        if (dry) {                                               // sizing pass: the blobs are not placed yet
            static const char some = 0;
            h1.wpack = h2.wpack = h1.zero_page = h2.zero_page = &some;
            h1.bias = h2.bias = (const float*)&some;
        }
        if (!demfi_resblock_eligible(&h1, &h2)) return;
        demfi_op op;
        memset(&op, 0, sizeof(op));
        op.kind = DEMFI_OP_RESBLOCK;
        op.conv = o1.conv;
        op.nch = o2.conv;
        op.macs = o1.macs + o2.macs;
        strncpy(op.name, name.c_str(), sizeof(op.name) - 1);
        seg.pop_back();
        seg.pop_back();
        seg.push_back(op);
        c->fused_now.push_back("resblock:" + name);
    }

    // Round 6: one SepConvGRU half-step (DeMFInet.py:844-849 / 851-856).  conv() has just appended the three plain 64-cout layers
    //     convr: [h, x] -> r*h (MUL)     convz: [h, x] -> z (sigmoid, into the z buffer)     convq: [r*h, x] -> h' (GRU epilogue, aux = z)
    // The first becomes a launch of the round-6 kernel's R mode, the other two ONE launch of its ZQ mode (gru.hip: z stays on chip, the z
    // buffer is never touched) when they qualify (fp16 plan).  The descriptors stay as they are: DEMFI_GRU6=0 and the CPU plan
    // interpreter run the three layers through demfi_conv2d (the round-5 kernel at 64 couts).
    void fuse_gru(OpList& seg, const std::string& name)
    {
        static const bool on = !(getenv("DEMFI_GRU6") && atoi(getenv("DEMFI_GRU6")) == 0);
        if (status < 0 || !on || seg.size() < 3) return;
        const demfi_op oq = seg[seg.size() - 1], oz = seg[seg.size() - 2], orr = seg[seg.size() - 3];
        if (oq.kind != DEMFI_OP_CONV || oz.kind != DEMFI_OP_CONV || orr.kind != DEMFI_OP_CONV) return;
        demfi_conv hq = c->descs[oq.conv], hz = c->descs[oz.conv], hr = c->descs[orr.conv];
        if (dry) {                                               // sizing pass: the blobs are not placed yet
            static const char some = 0;
            for (demfi_conv* h : {&hq, &hz, &hr}) { h->wpack = h->zero_page = &some; h->bias = (const float*)&some; }
        }
        if (!demfi_gru_r_eligible(&hr) || !demfi_gru_zq_eligible(&hz, &hq)) return;
        demfi_op r = orr, zq;
        r.kind = DEMFI_OP_GRU_R;
        memset(&zq, 0, sizeof(zq));
        zq.kind = DEMFI_OP_GRU_ZQ;
        zq.conv = oz.conv;
        zq.nch = oq.conv;
        zq.macs = oz.macs + oq.macs;
        strncpy(zq.name, (name + ".convzq").c_str(), sizeof(zq.name) - 1);
        seg.pop_back(); seg.pop_back(); seg.pop_back();
        seg.push_back(r);
        seg.push_back(zq);
        c->fused_now.push_back("gru:" + name);
    }

    // x_{k+1} = x_k + conv2(relu(conv1(x_k))) ping-ponging between buffers a and b (t = scratch); returns the result buffer
    const Tensor* resblocks(OpList& seg, const std::string& prefix, int n, const Tensor& a, const Tensor& t, const Tensor& b,
                            int H, int W, int batch)
    {
        const Tensor *cur = &a, *other = &b;
        for (int i = 0; i < n; ++i) {
            const std::string p = prefix + "." + std::to_string(i);
            conv(seg, p + ".conv1", {fsrc(*cur, 0)}, {D(fview(t), range(0, 64), DEMFI_ACT_RELU)}, H, W, 1, batch);
            conv(seg, p + ".conv2", {fsrc(t, 0)}, {D(fview(*other), range(0, 64), DEMFI_ACT_NONE, DEMFI_MODE_STORE, fview(*cur))},
                 H, W, 1, batch);
            fuse_resblock(seg, p);
            std::swap(cur, other);
        }
        return cur;
    }

    // ---- single-call operators: the launch sequences demfi_amd/ops.py composes, behind the C ABI --------------------
    void build_operator()
    {
        BufSet& B = c->tr_bufs[0];
        OpList& tr = c->tr_ops[0];
        const int H = c->H, W = c->W, nb = c->op_batch;
        const int R = DEMFI_ACT_RELU, S = DEMFI_ACT_SIGMOID;
        if (c->op_kind == 1) {
            // SepConvGRU.forward (DeMFInet.py:838-857): horizontal then vertical GRU step; z | r as one 128-cout convolution
            // (sigmoid; sigmoid * h), q with the GRU blend (1 - z) h + z tanh(.) in its epilogue
            const Tensor* h = &B["h"];
            for (int s2 = 0; s2 < 2 && status >= 0; ++s2) {
                const std::string sfx = std::to_string(s2 + 1);
                std::vector<float> zw, zb;
                Layer shape = {128, 128, s2 == 0 ? 1 : 5, s2 == 0 ? 5 : 1};
                for (const char* g : {"z", "r"}) {
                    auto iw = c->weights.find(std::string("conv") + g + sfx + ".weight"), ib = c->weights.find(std::string("conv") + g + sfx + ".bias");
                    if (dry) continue;
                    if (iw == c->weights.end() || ib == c->weights.end()) { status = demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: GRU weights were not loaded"); return; }
                    zw.insert(zw.end(), iw->second.data.begin(), iw->second.data.end());
                    zb.insert(zb.end(), ib->second.data.begin(), ib->second.data.end());
                }
                const Tensor& hnext = s2 == 0 ? B["h1"] : B["out"];
                static const bool gru6_env = !(getenv("DEMFI_GRU6") && atoi(getenv("DEMFI_GRU6")) == 0);
                if (gru6_env && c->dtype == DEMFI_F16) {         // round 6: r*h, then z + q + blend in one launch (fuse_gru)
                    conv(tr, "convr" + sfx, {fsrc(*h, 0), fsrc(B["x"], 64)}, {D(fview(B["rh"]), range(0, 64), DEMFI_ACT_NONE, DEMFI_MODE_MUL, fview(*h))}, H, W, 1, nb);
                    conv(tr, "convz" + sfx, {fsrc(*h, 0), fsrc(B["x"], 64)}, {D(fview(B["z"]), range(0, 64), S)}, H, W, 1, nb);
                    conv(tr, "convq" + sfx, {fsrc(B["rh"], 0), fsrc(B["x"], 64)},
                         {D(fview(hnext), range(0, 64), DEMFI_ACT_NONE, DEMFI_MODE_GRU, fview(*h), fview(B["z"]))}, H, W, 1, nb);
                    fuse_gru(tr, "step" + sfx);
                    h = &hnext;
                    continue;
                }
                conv(tr, "convzr" + sfx, {fsrc(*h, 0), fsrc(B["x"], 64)},
                     {D(fview(B["z"]), range(0, 64), S), D(fview(B["rh"]), range(64, 128), DEMFI_ACT_NONE, DEMFI_MODE_MUL, fview(*h))}, H, W, 1, nb,
                     &zw, &zb, &shape);
                conv(tr, "convq" + sfx, {fsrc(B["rh"], 0), fsrc(B["x"], 64)},
                     {D(fview(hnext), range(0, 64), DEMFI_ACT_NONE, DEMFI_MODE_GRU, fview(*h), fview(B["z"]))}, H, W, 1, nb);
                h = &hnext;
            }
            return;
        }
        // FGAC.forward at rr = sr = 0 (DeMFInet.py:386-452): conv_ref_k -> bilinear sample at the absolute flow coordinates -> fusion ->
        // w = sigmoid(w_gen_2(relu(w_gen(cat[source, E_s])))) -> w source + (1 - w) E_s
        const int64_t hw4 = (int64_t)H * W * 4;
        conv(tr, "conv_ref_k", {fsrc(B["ref"], 0)}, {D(fview(B["ref_k"]), range(0, 64))}, H, W, 1, nb);
        for (int b = 0; b < nb; ++b) {
            demfi_op o = blank();
            o.nch = 64;
            o.a = fview(B["ref_k"], 0, b); o.o = fview(B["sampled"], 0, b);
            o.p[0] = ptr(B["flow"]) + 2 * b * hw4;
            simple(tr, DEMFI_OP_FGAC, "fgac", o);
        }
        conv(tr, "fusion", {fsrc(B["sampled"], 0)}, {D(fview(B["e_s"]), range(0, 64))}, H, W, 1, nb);
        conv(tr, "w_gen", {fsrc(B["source"], 0), fsrc(B["e_s"], 64)}, {D(fview(B["hid"]), range(0, 64), R)}, H, W, 1, nb);
        conv(tr, "w_gen_2", {fsrc(B["hid"], 0)}, {D(tview(B["w"], 0, (int64_t)H * W), {0}, S)}, H, W, 1, nb);
        for (int b = 0; b < nb; ++b) {
            demfi_op o = blank();
            o.nch = 64;
            o.a = fview(B["source"], 0, b); o.b = fview(B["e_s"], 0, b); o.o = fview(B["out"], 0, b);
            o.p[0] = ptr(B["w"]) + b * hw4;
            simple(tr, DEMFI_OP_GATE, "gate", o);
        }
    }

    void build_trunk(int k)
    {
        BufSet& B = c->tr_bufs[k];
        OpList& tr = c->tr_ops[k];
        const int H = c->H, W = c->W, H2 = H / 2, W2 = W / 2;
        const int R = DEMFI_ACT_RELU, T = DEMFI_ACT_TANH, S = DEMFI_ACT_SIGMOID;
        const int64_t hw4 = (int64_t)H * W * 4;
        // ============================ trunk: FF_RDB (DeMFInet.py:233-253) ==========================================
        std::string p = "FF_RDB_Module.";
        { demfi_op o = blank(); o.p[0] = ptr(B["x"]); o.p[1] = ptr(B["s2d"]); simple(tr, DEMFI_OP_S2D, "s2d", o); }
        { demfi_op o = blank(); o.p[0] = ptr(B["x"]); o.p[1] = ptr(B["overlay"]); simple(tr, DEMFI_OP_OVERLAY, "overlay", o); }
        conv(tr, p + "SFENet1", {fsrc(B["s2d"], 0)}, {D(fview(B["f1"]), range(0, 96))}, H2, W2);
        conv(tr, p + "SFENet2", {fsrc(B["f1"], 0)}, {D(fview(B["x0"]), range(0, 96))}, H2, W2);
        for (int i = 0; i < 12; ++i) {
            auto xin = [&]() { return i == 0 ? fsrc(B["x0"], 0) : fsrc(B["gffcat"], 0, 96 * (i - 1), 96); };
            const demfi_view xres = i == 0 ? fview(B["x0"]) : fview(B["gffcat"], 96 * (i - 1));
            const std::string rp = p + "RDBs." + std::to_string(i);
            for (int q = 0; q < 4; ++q) {
                std::vector<Src> s{xin()};
                if (q) s.push_back(fsrc(B["grow"], 96, 0, 32 * q));
                conv(tr, rp + ".convs." + std::to_string(q) + ".conv.0", s, {D(fview(B["grow"], 32 * q), range(0, 32), R)}, H2, W2);
            }
            conv(tr, rp + ".LFF", {xin(), fsrc(B["grow"], 96, 0, 128)},
                 {D(fview(B["gffcat"], 96 * i), range(0, 96), DEMFI_ACT_NONE, DEMFI_MODE_STORE, xres)}, H2, W2);
        }
        conv(tr, p + "GFF.0", {fsrc(B["gffcat"], 0)}, {D(fview(B["g0"]), range(0, 96))}, H2, W2);
        conv(tr, p + "GFF.1", {fsrc(B["g0"], 0)}, {D(fview(B["g1"]), range(0, 96), DEMFI_ACT_NONE, DEMFI_MODE_STORE, fview(B["f1"]))}, H2, W2);
        // UPNet.0 + PixelShuffle(2): out[c, 2h+i, 2w+j] = conv[c*4 + i*2 + j, h, w]
        {
            std::vector<Dst> ds;
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) {
                    std::vector<int32_t> co;
                    for (int ch = 0; ch < 64; ++ch) co.push_back(ch * 4 + i * 2 + j);
                    ds.push_back(D(fview(B["up"]), co, DEMFI_ACT_NONE, DEMFI_MODE_STORE, NOVIEW, NOVIEW, 2, i, j));
                }
            conv(tr, p + "UPNet.0", {fsrc(B["g1"], 0)}, ds, H2, W2);
        }
        // UPNet.2 (3x3, 64 -> 133 = F0 | F1 | flow_01, flow_10, occlusion logit; DeMFInet.py:231, 247-253).  Round 6, fp16 plan: one launch per
        // output group -- the two tanh feature halves on the staged-store 64 -> 64 kernel, the 5 planes on the thin-output kernel -- instead
        // of ONE 160-cout launch of the general kernel (0.32 ms at 0.18 of the matrix peak).  DEMFI_UP2_SPLIT=0: the single launch.
        static const bool up2_split = !(getenv("DEMFI_UP2_SPLIT") && atoi(getenv("DEMFI_UP2_SPLIT")) == 0);
        if (c->dtype == DEMFI_F16 && up2_split) {
            SubW w0 = sub_weight_cout(p + "UPNet.2", 0, 64), w1 = sub_weight_cout(p + "UPNet.2", 64, 64), w2 = sub_weight_cout(p + "UPNet.2", 128, 5);
            conv(tr, p + "UPNet.2#F0", {fsrc(B["up"], 0)}, {D(fview(B["F01"], 0, 0), range(0, 64), T)}, H, W, 1, 1, &w0.w, &w0.b, &w0.shape);
            conv(tr, p + "UPNet.2#F1", {fsrc(B["up"], 0)}, {D(fview(B["F01"], 0, 1), range(0, 64), T)}, H, W, 1, 1, &w1.w, &w1.b, &w1.shape);
            conv(tr, p + "UPNet.2#f", {fsrc(B["up"], 0)}, {D(tview(B["ffo"]), range(0, 5))}, H, W, 1, 1, &w2.w, &w2.b, &w2.shape);
        } else
        conv(tr, p + "UPNet.2", {fsrc(B["up"], 0)},
             {D(fview(B["F01"], 0, 0), range(0, 64), T), D(fview(B["F01"], 0, 1), range(64, 128), T), D(tview(B["ffo"]), range(128, 133))}, H, W);
        // ============================ trunk: FAC-FB (DeMFInet.py:335-358, 386-452) ================================
        p = "FAC_FB_Module.";
        conv(tr, p + "conv_first", {fsrc(B["F01"], 0)}, {D(fview(B["enc_a"]), range(0, 64), R)}, H, W, 1, 2);
        const Tensor* enc = resblocks(tr, p + "feature_extraction", c->hp.num_resb_facfb, B["enc_a"], B["enc_t"], B["enc_b"], H, W, 2);
        B["enc"] = *enc;                                             // alias: the buffer holding the encoder output
        for (int b = 0; b < 2; ++b) {          // b = 0: F1 -> F0 with flow_01 ; b = 1: F0 -> F1 with flow_10 (346-349)
            const std::string fg = p + (c->hp.shared_fgac ? "shared_FGAC" : (b == 0 ? "FGAC_F1toF0" : "FGAC_F0toF1"));
            const int ref = 1 - b, src = b;
            conv(tr, fg + ".conv_ref_k", {fsrc(*enc, 0, 0, -1, ref)}, {D(fview(B["rk"], 0, b), range(0, 64))}, H, W);
            if (c->hp.fgac_rr > 0) {
                // generalised FGAC (DeMFInet.py:401-445): conv_source_k is live, optional PxP average pooling of both key
                // maps, then the window kernel (correlation, softmax, weighted sum)
                conv(tr, fg + ".conv_source_k", {fsrc(*enc, 0, 0, -1, src)}, {D(fview(B["skk"], 0, b), range(0, 64))}, H, W);
                const char *rkn = "rk", *skn = "skk";
                if (c->hp.fgac_sr > 0) {
                    for (int q = 0; q < 2; ++q) {
                        demfi_op o = blank();
                        o.nch = 64; o.conv = c->hp.fgac_sr;
                        o.a = fview(B[q ? "skk" : "rk"], 0, b); o.o = fview(B[q ? "skp" : "rkp"], 0, b);
                        simple(tr, DEMFI_OP_AVG_POOL, "avg_pool", o);
                    }
                    rkn = "rkp"; skn = "skp";
                }
                demfi_op o = blank();
                o.nch = 64; o.conv = c->hp.fgac_rr; o._pad = c->hp.flags & DEMFI_HP_FGAC_CENTRED;
                o.a = fview(B[rkn], 0, b); o.b = fview(B[skn], 0, b); o.o = fview(B["smp"], 0, b);
                o.p[0] = ptr(B["ffo"]) + (b == 0 ? 0 : 2) * hw4;
                simple(tr, DEMFI_OP_FGAC_WINDOW, "fgac_window", o);
            } else {
                demfi_op o = blank();
                o.nch = 64;
                o.a = fview(B["rk"], 0, b); o.o = fview(B["smp"], 0, b);
                o.p[0] = ptr(B["ffo"]) + (b == 0 ? 0 : 2) * hw4;
                simple(tr, DEMFI_OP_FGAC, "fgac", o);
            }
            conv(tr, fg + ".fusion", {fsrc(B["smp"], 0, 0, -1, b)}, {D(fview(B["E"], 0, b), range(0, 64))}, H, W);
            conv(tr, fg + ".w_gen", {fsrc(*enc, 0, 0, -1, src), fsrc(B["E"], 64, 0, -1, b)}, {D(fview(B["wg"], 0, b), range(0, 64), R)}, H, W);
            conv(tr, fg + ".w_gen_2", {fsrc(B["wg"], 0, 0, -1, b)}, {D(tview(B["gate"], b), {0}, S)}, H, W);
            {
                demfi_op o = blank();
                o.nch = 64;
                o.a = fview(*enc, 0, b); o.b = fview(B["E"], 0, b); o.o = fview(B["aF"], 0, b);
                o.p[0] = ptr(B["gate"]) + b * hw4;
                simple(tr, DEMFI_OP_GATE, "gate", o);
            }
            if (c->hp.flags & DEMFI_HP_EXTRAS) {
                // the maps FGAC.forward returns besides its output (DeMFInet.py:454-496): diff (always computed by the reference, returned
                // in the training / visualisation tuples of DeMFInet.forward 167-176) and the four visualisation maps + (1 - w_sr)
                auto vz = [&](int k) { return ptr(B["viz"]) + (6 * b + k) * hw4; };
                auto absmean = [&](int k, demfi_view a, demfi_view bb) {
                    demfi_op o = blank();
                    o.conv = 0; o.nch = 64; o.a = a; o.b = bb; o.p[0] = vz(k);
                    simple(tr, DEMFI_OP_VIZ, "viz_absmean", o);
                    demfi_op n = blank();
                    n.conv = 1; n.p[0] = vz(k); n.p[1] = ptr(B["vizs"]);
                    simple(tr, DEMFI_OP_VIZ, "viz_normalize", n);
                };
                { demfi_op o = blank(); o.conv = 2; o.p[0] = vz(0); o.p[1] = ptr(B["gate"]) + b * hw4; simple(tr, DEMFI_OP_VIZ, "viz_one_minus", o); }
                absmean(1, fview(*enc, 0, src), NOVIEW);                    // source_v
                absmean(2, fview(B["rk"], 0, b), NOVIEW);                   // init_ref_k = conv_ref_k(ref)
                absmean(3, fview(B["E"], 0, b), NOVIEW);                    // E_s
                absmean(4, fview(B["aF"], 0, b), NOVIEW);                   // bolstered_F_s
                absmean(5, fview(B["aF"], 0, b), fview(*enc, 0, src));      // diff = bolstered_F_s - source_v
            }
        }
        if (c->dtype == DEMFI_F16) {
            // Refine_Module.enc1 = conv4x4s2(cat[aF0, aF1 | Ft, flows ...]) (DeMFInet.py:77, 588): the aF0 | aF1 half (128 of
            // 201 input channels, 64 % of the layer) does not depend on t -> computed once per window, added as a residual
            SubW wa = sub_weight("Refine_Module.enc1", range(0, 128), true);
            conv(tr, "Refine_Module.enc1#aF", {fsrc(B["aF"], 0, 0, -1, 0), fsrc(B["aF"], 64, 0, -1, 1)},
                 {D(fview(B["u1a"]), range(0, 64))}, H2, W2, 2, 1, &wa.w, &wa.b, &wa.shape);
            // Mixer.conv_ref1 (7x7 over 30 planes) and Dec_first_2 read the 4 input frames and flow_10 | flow_01: 16 planes that do
            // not change within a window.  Packed once (xff16) and their share of both layers computed once per window; the
            // per-t parts then fit the narrow persistent kernels (16-channel records) and take these as residuals.
            std::vector<const float*> pl;
            for (int f = 0; f < 4; ++f)
                for (int col = 0; col < 3; ++col) pl.push_back(plane(B["x"], col * 4 + f));
            for (int i : {2, 3, 0, 1}) pl.push_back(plane(B["ffo"], i));
            pack(tr, pl, B["xff16"]);
            SubW w1 = sub_weight("Booster_Module.Mixer.conv_ref1", range(9, 25), false);
            conv(tr, "Booster_Module.Mixer.conv_ref1#win", {fsrc_map(B["xff16"], range(0, 16))}, {D(fview(B["re1w"]), range(0, 32))}, H, W, 1, 1,
                 &w1.w, &w1.b, &w1.shape);
            std::vector<int32_t> sel = range(87, 99);                    // frames, then flow_10 | flow_01 (Agg3 order, DeMFInet.py:151-155)
            for (int i = 78; i < 82; ++i) sel.push_back(i);
            SubW w2 = sub_weight("Dec_first_2", sel, false);
            conv(tr, "Dec_first_2#win", {fsrc_map(B["xff16"], range(0, 16))}, {D(fview(B["g_pw"]), range(0, 64))}, H, W, 1, 1, &w2.w, &w2.b,
                 &w2.shape);
        }
    }

    void warp(OpList& seg, const char* name, int C, demfi_view A, demfi_view Bv, demfi_view O, const void* fa, const void* fb,
              const void* logit, const void* occ_out, const void* t, const void* pack8 = nullptr)
    {
        demfi_op o = blank();
        o.nch = C; o.a = A; o.b = Bv; o.o = O;
        o.p[0] = fa; o.p[1] = fb; o.p[2] = logit; o.p[3] = occ_out; o.p[4] = pack8; o.t = t;
        simple(seg, DEMFI_OP_WARP, name, o);
    }

    void build_t(int k, int q)
    {
        BufSet& TB = c->tr_bufs[k];
        BufSet& B = c->t_bufs[k][q];
        OpList& th = tb > 1 ? c->tb_head_ops[k] : c->head_ops[k][q];
        const int H = c->H, W = c->W, N = c->N;
        const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
        const int R = DEMFI_ACT_RELU, T = DEMFI_ACT_TANH;
        const int64_t hw4 = (int64_t)H * W * 4;
        const Tensor& ffo = TB["ffo"];
        const Tensor& aF = TB["aF"];
        const Tensor& x = TB["x"];
        const void* tp = ptr(B["t"]);
        static const bool cfr_pack = !(getenv("DEMFI_CFR_PACK") && atoi(getenv("DEMFI_CFR_PACK")) == 0);
        auto delta_v = [&](int step, int ch) { return tview(B["delta"], 5 * step + ch); };
        auto delta_p = [&](int step, int ch) { return plane(B["delta"], 5 * step + ch); };
        // ============================ per-t head: CFR, FWB, refinement, D1, Ch_Reducer ==============================
        {
            demfi_op o = blank();
            o.p[0] = ptr(ffo); o.p[1] = ptr(ffo) + 2 * hw4; o.p[2] = ptr(B["cfr_acc"]); o.p[3] = ptr(B["ft"]); o.t = tp;
            // round 6: the finish also writes misc16 = [flow_t0, flow_t1 | flow_01, flow_10, occ logit | 0] (the thin members of Agg1, 77) as the
            // NHWC record enc1 stages: one plane-pack launch per window less (DEMFI_CFR_PACK=0: the pack launch of rounds 1-5)
            if (cfr_pack) { o.p[4] = ptr(ffo) + 4 * hw4; o.p[5] = ptr(B["misc16"]); }
            simple(th, DEMFI_OP_CFR, "cfr", o);
        }
        warp(th, "warp_fat", 64, fview(TB["F01"], 0, 0), fview(TB["F01"], 0, 1), fview(B["Ft"], 0, 0), ptr(B["ft"]), ptr(B["ft"]) + 2 * hw4,
             ptr(ffo) + 4 * hw4, nullptr, tp);
        std::string p = "Refine_Module.";
        // Agg1 = cat[aF0, aF1, Ft, flow_t0, flow_t1, flow_01, flow_10, occ_0_logit] (DeMFInet.py:77)
        if (!cfr_pack) {
            std::vector<const float*> pl;
            for (int i = 0; i < 4; ++i) pl.push_back(plane(B["ft"], i));
            for (int i = 0; i < 5; ++i) pl.push_back(plane(ffo, i));
            pack(th, pl, B["misc16"]);
        }
        {
            std::vector<int32_t> m = range(192, 201);
            m.insert(m.end(), 7, -1);
            if (c->dtype == DEMFI_F16) {
                // the t-dependent 73 channels only; + the hoisted aF part (trunk) as residual, then ReLU
                std::vector<int32_t> sel = range(128, 201), m2 = range(64, 73);
                m2.insert(m2.end(), 7, -1);
                SubW wb = sub_weight(p + "enc1", sel, false);
                conv(th, p + "enc1#t", {fsrc(B["Ft"], 0), fsrc_map(B["misc16"], m2)},
                     {D(fview(B["u1"]), range(0, 64), R, DEMFI_MODE_STORE, fview(TB["u1a"]))}, H2, W2, 2, 1, &wb.w, &wb.b, &wb.shape);
            } else
            conv(th, p + "enc1", {fsrc(aF, 0, 0, -1, 0), fsrc(aF, 64, 0, -1, 1), fsrc(B["Ft"], 128), fsrc_map(B["misc16"], m)},
                 {D(fview(B["u1"]), range(0, 64), R)}, H2, W2, 2);
        }
        conv(th, p + "enc2", {fsrc(B["u1"], 0)}, {D(fview(B["u2"]), range(0, 128), R)}, H4, W4, 2);
        conv(th, p + "enc3", {fsrc(B["u2"], 0)}, {D(fview(B["u3"]), range(0, 256), R)}, H8, W8, 2);
        conv(th, p + "dec0", {fsrc(B["u3"], 0)}, {D(fview(B["d0"]), range(0, 256), R)}, H8, W8);
        conv(th, p + "dec1", {fsrc(B["d0"], 0, 0, -1, -1, 1), fsrc(B["u2"], 256)}, {D(fview(B["d1"]), range(0, 128), R)}, H4, W4);
        conv(th, p + "dec2", {fsrc(B["d1"], 0, 0, -1, -1, 1), fsrc(B["u1"], 128)}, {D(fview(B["d2"]), range(0, 64), R)}, H2, W2);
        // + cat[flow_t0, flow_t1, occ_0_logit, aF0, aF1] (78-80), tanh on the feature part (86-87)
        if (c->dtype == DEMFI_F16) {
            // dec3 = conv3x3(NN-upsample x2 (d2)) (DeMFInet.py:600-602).  A 3x3 filter over a 2x nearest-neighbour upsampled image
            // reads, for the output pixels of parity (dy, dx), only 2x2 DISTINCT low-resolution pixels: rows {y-1, y} with weights
            // {W[0], W[1]+W[2]} for dy = 0, rows {y, y+1} with {W[0]+W[1], W[2]} for dy = 1 (columns alike).  Four 2x2 convolutions
            // on the half-resolution grid, each writing its parity of the full-resolution outputs (views with doubled strides),
            // do 4 taps per output instead of 9: 2.25x fewer MACs, no upsampled gather.
            auto it = c->table.find(p + "dec3");
            const Layer l3 = it->second;
            auto iw = c->weights.find(p + "dec3.weight"), ib = c->weights.find(p + "dec3.bias");
            if (!dry && (iw == c->weights.end() || ib == c->weights.end())) {
                status = demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: weight '%s' was not loaded", (p + "dec3").c_str());
                return;
            }
            auto phase_view = [&](demfi_view v, int dy, int dx) {           // pixels (2y+dy, 2x+dx) of a full-resolution view
                const int64_t elt = v.is_f32 ? 4 : 2;
                v.ptr = (char*)v.ptr + ((int64_t)dy * v.sy + (int64_t)dx * v.sx) * elt;
                v.sx *= 2; v.sy *= 2;
                return v;
            };
            // Round 3: each parity runs as THREE launches of the fast 64-input-channel kernels instead of one 160-cout launch of the
            // general kernel (0.50 ms per parity at 0.09 of the MFMA peak): the 2x2 filter is embedded in a 3x3 one (taps outside the
            // 2x2 footprint are zero: the 2.25x MAC saving is given back, the layer is memory-bound either way), so that the two
            // 64-channel feature halves (tanh + residual aF0 / aF1) go to the staged-store 64 -> 64 kernel and the 5 flow / occlusion
            // planes to a 32-cout launch.
            const Layer l9 = {l3.cout, l3.cin, 3, 3};
            // round 6: the flow / occlusion planes of the two column parities of a row parity in ONE launch (10 couts = 4 live octets of a
            // 32-cout subtile that the per-parity launches filled with 5): 4 -> 2 launches of the thin kernel (DEMFI_DEC3F_PAIR=0: one per parity)
            static const bool f_pair = !(getenv("DEMFI_DEC3F_PAIR") && atoi(getenv("DEMFI_DEC3F_PAIR")) == 0);
            for (int dy = 0; dy < 2 && status >= 0; ++dy) {
                std::vector<float> wf2, bf2;
                for (int dx = 0; dx < 2 && status >= 0; ++dx) {
                    std::vector<float> w9e, b2;
                    if (!dry) {
                        w9e.assign((size_t)l3.cout * l3.cin * 9, 0.0f);
                        for (int co = 0; co < l3.cout; ++co)
                            for (int ci = 0; ci < l3.cin; ++ci) {
                                const float* w9 = &iw->second.data[((size_t)co * l3.cin + ci) * 9];
                                float* we = &w9e[((size_t)co * l3.cin + ci) * 9];
                                for (int ky = 0; ky < 3; ++ky)
                                    for (int kx = 0; kx < 3; ++kx) {
                                        // source row of tap ky for output parity dy: rows {y-1, y} (dy = 0) or {y, y+1} (dy = 1) of the low-res image
                                        const int a = dy == 0 ? (ky >= 1) : (ky >= 2), b = dx == 0 ? (kx >= 1) : (kx >= 2);
                                        we[(a + dy) * 3 + (b + dx)] += w9[ky * 3 + kx];      // embedded position: low-res row y - 1 + (a + dy)
                                    }
                            }
                        b2 = ib->second.data;
                    }
                    const std::string nm = p + "dec3#p" + std::to_string(dy) + std::to_string(dx);
                    // every output channel of a call's weight must be routed: slice the embedded filter per launch
                    auto slice = [&](int c0, int c1, std::vector<float>& w, std::vector<float>& b) {
                        if (dry) return;
                        w.assign(w9e.begin() + (size_t)c0 * l3.cin * 9, w9e.begin() + (size_t)c1 * l3.cin * 9);
                        b.assign(b2.begin() + c0, b2.begin() + c1);
                    };
                    std::vector<float> wa, ba, wb, bb, wf, bf;
                    slice(5, 69, wa, ba); slice(69, 133, wb, bb); slice(0, 5, wf, bf);
                    const Layer l64 = {64, l3.cin, 3, 3}, l5 = {5, l3.cin, 3, 3};
                    conv(th, nm + "a", {fsrc(B["d2"], 0)},
                         {D(phase_view(fview(B["rF"], 0, 0), dy, dx), range(0, 64), T, DEMFI_MODE_STORE, phase_view(fview(aF, 0, 0), dy, dx))},
                         H2, W2, 1, 1, &wa, &ba, &l64);
                    conv(th, nm + "b", {fsrc(B["d2"], 0)},
                         {D(phase_view(fview(B["rF"], 0, 1), dy, dx), range(0, 64), T, DEMFI_MODE_STORE, phase_view(fview(aF, 0, 1), dy, dx))},
                         H2, W2, 1, 1, &wb, &bb, &l64);
                    if (f_pair) {
                        wf2.insert(wf2.end(), wf.begin(), wf.end());
                        bf2.insert(bf2.end(), bf.begin(), bf.end());
                        continue;
                    }
                    // the 5 planes also go, as fp16, into the record Mixer.conv_delta1 stages (delta16): no plane-packing launch
                    pack_next(phase_view(fview(B["delta16"]), dy, dx), 0, 4);
                    conv(th, nm + "f", {fsrc(B["d2"], 0)},
                         {D(phase_view(delta_v(0, 0), dy, dx), range(0, 4), DEMFI_ACT_NONE, DEMFI_MODE_STORE, phase_view(tview(B["ft"]), dy, dx)),
                          D(phase_view(delta_v(0, 4), dy, dx), {4}, DEMFI_ACT_NONE, DEMFI_MODE_STORE, phase_view(tview(ffo, 4), dy, dx))},
                         H2, W2, 1, 1, &wf, &bf, &l5);
                }
                if (f_pair && status >= 0) {
                    // couts 0..4: column parity 0, 5..9: column parity 1; the packed copy of parity 1 lies one pixel (16 channels) further in delta16
                    const Layer l10 = {10, l3.cin, 3, 3};
                    pack_next(phase_view(fview(B["delta16"]), dy, 0), 0, 4, 16, 20);
                    conv(th, p + "dec3#p" + std::to_string(dy) + "xf", {fsrc(B["d2"], 0)},
                         {D(phase_view(delta_v(0, 0), dy, 0), range(0, 4), DEMFI_ACT_NONE, DEMFI_MODE_STORE, phase_view(tview(B["ft"]), dy, 0)),
                          D(phase_view(delta_v(0, 4), dy, 0), {4}, DEMFI_ACT_NONE, DEMFI_MODE_STORE, phase_view(tview(ffo, 4), dy, 0)),
                          D(phase_view(delta_v(0, 0), dy, 1), range(5, 9), DEMFI_ACT_NONE, DEMFI_MODE_STORE, phase_view(tview(B["ft"]), dy, 1)),
                          D(phase_view(delta_v(0, 4), dy, 1), {9}, DEMFI_ACT_NONE, DEMFI_MODE_STORE, phase_view(tview(ffo, 4), dy, 1))},
                         H2, W2, 1, 1, &wf2, &bf2, &l10);
                }
            }
        } else
        conv(th, p + "dec3", {fsrc(B["d2"], 0, 0, -1, -1, 1)},
             {D(fview(B["rF"], 0, 0), range(5, 69), T, DEMFI_MODE_STORE, fview(aF, 0, 0)),
              D(fview(B["rF"], 0, 1), range(69, 133), T, DEMFI_MODE_STORE, fview(aF, 0, 1)),
              D(delta_v(0, 0), range(0, 4), DEMFI_ACT_NONE, DEMFI_MODE_STORE, tview(B["ft"])),
              D(delta_v(0, 4), {4}, DEMFI_ACT_NONE, DEMFI_MODE_STORE, tview(ffo, 4))}, H, W);
        warp(th, "warp_fat", 64, fview(B["rF"], 0, 0), fview(B["rF"], 0, 1), fview(B["rF"], 0, 2), delta_p(0, 0), delta_p(0, 2),
             delta_p(0, 4), plane(B["occ"], 0), tp);                 // rFt -> rF[2], occ[0]
        // D1 on the three frames (Conv3d depth = batch), DeMFInet.py:95-101
        conv(th, "Dec_first", {fsrc(B["rF"], 0)}, {D(fview(B["dec_a"]), range(0, 64), R)}, H, W, 1, 3);
        const Tensor* cur = resblocks(th, "Decoder_res", c->hp.num_resb_dec, B["dec_a"], B["dec_t"], B["dec_b"], H, W, 3);
        conv(th, "Dec_last1", {fsrc(*cur, 0)}, {D(fview(B["dec_t"]), range(0, 64), R)}, H, W, 1, 3);
        conv(th, "Dec_last2", {fsrc(B["dec_t"], 0)}, {D(tview(B["sharp1"], 0, 3ll * H * W), range(0, 3))}, H, W, 1, 3);
        conv(th, "Ch_Reducer", {fsrc(B["rF"], 0, 0, -1, 0), fsrc(B["rF"], 64, 0, -1, 1), fsrc(B["rF"], 128, 0, -1, 2)},
             {D(fview(B["frec0"]), range(0, 64), T)}, H, W);
        // Mixer reference branch (iteration-invariant, hoisted): cat[S0p,S1p,Stp,B0,B1,B-1,B2 | flow_10,flow_01 | t_ref]
        p = "Booster_Module.";
        std::vector<const float*> xpl;                              // B0, B1, B-1, B2 colour planes (cat order)
        for (int f = 0; f < 4; ++f)
            for (int col = 0; col < 3; ++col) xpl.push_back(plane(x, col * 4 + f));
        std::vector<int32_t> agg3s_cin = range(0, 6);                // fp32 plan: channel map of the 27-plane pack
        if (c->dtype == DEMFI_F16) {
            // per-t planes only; the window-constant 16 planes were done in the trunk (xff16 -> re1w, g_pw)
            std::vector<const float*> pl;
            for (int i = 0; i < 9; ++i) pl.push_back(plane(B["sharp1"], i));
            for (int i = 0; i < 5; ++i) pl.push_back(delta_p(0, i));
            // round 6: channel 14 = occ_0, so that this ONE record also serves Dec_first_2's recursion-invariant per-t planes (rounds 2-5
            // packed a second record, agg16 = S0p, S1p | occ_0 | rflow, from the same planes: one more launch per window, 0.29 ms)
            pl.push_back(plane(B["occ"], 0));
            pack(th, pl, B["ref16"]);
            std::vector<int32_t> sel = range(0, 9), m = range(0, 14);
            for (int i = 25; i < 30; ++i) sel.push_back(i);
            m.insert(m.end(), 2, -1);
            SubW w1 = sub_weight(p + "Mixer.conv_ref1", sel, true);
            conv(th, p + "Mixer.conv_ref1#t", {fsrc_map(B["ref16"], m)}, {D(fview(B["re1"]), range(0, 32), R, DEMFI_MODE_STORE, fview(TB["re1w"]))},
                 H, W, 1, 1, &w1.w, &w1.b, &w1.shape);
            // the iteration-invariant, t-dependent part of Agg3 (DeMFInet.py:151-155: S0p,S1p | occ_0 | rflow_t0,t1) is read from ref16
            // through a channel map (dyn_m16 below)
        } else {
            {
                std::vector<const float*> pl;
                for (int i = 0; i < 9; ++i) pl.push_back(plane(B["sharp1"], i));
                pl.insert(pl.end(), xpl.begin(), xpl.end());
                for (int i : {2, 3, 0, 1}) pl.push_back(plane(ffo, i));
                for (int i = 0; i < 5; ++i) pl.push_back(delta_p(0, i));
                pack(th, pl, B["ref32"]);
                std::vector<int32_t> m = range(0, 30);
                m.insert(m.end(), 2, -1);
                conv(th, p + "Mixer.conv_ref1", {fsrc_map(B["ref32"], m)}, {D(fview(B["re1"]), range(0, 32), R)}, H, W);
            }
            // iteration-invariant part of Agg3 (DeMFInet.py:151-155): S0p,S1p | occ_0 | rflow_t0,t1 | flow_10,flow_01 | frames
            std::vector<const float*> pl;
            for (int i = 0; i < 6; ++i) pl.push_back(plane(B["sharp1"], i));
            pl.push_back(plane(B["occ"], 0));
            for (int i = 0; i < 4; ++i) pl.push_back(delta_p(0, i));
            for (int i : {2, 3, 0, 1}) pl.push_back(plane(ffo, i));
            pl.insert(pl.end(), xpl.begin(), xpl.end());
            pack(th, pl, B["agg3s"]);
            agg3s_cin.push_back(73);
            for (int i = 74; i < 82; ++i) agg3s_cin.push_back(i);
            for (int i = 87; i < 99; ++i) agg3s_cin.push_back(i);
            agg3s_cin.insert(agg3s_cin.end(), 5, -1);
        }
        conv(th, p + "Mixer.conv_ref2", {fsrc(B["re1"], 0)}, {D(fview(B["rd64"], 0), range(0, 32), R)}, H, W);
        // Dec_first_2 = relu(conv3x3(Agg3)) with Agg3 = cat[F_rec (64, changes per recursion) | 27 recursion-invariant planes |
        // 8 planes of the current recursion] (DeMFInet.py:151-157), split by linearity in the fp16 plan (see below).
        const std::vector<int32_t> a3d_sel = {6, 7, 8, 82, 83, 84, 85, 86};
        SubW w_dyn, w_rec, w_df2;
        std::vector<int32_t> dyn_m16 = range(0, 11), df2_m16, df2_m8;
        // round 6 EXPERIMENT (DEMFI_DF2_FUSE=1; the product keeps the two launches of rounds 2-5): ONE launch per recursion on the
        // streamed-weight kernel (wsconv.hip): units [F_rec lo | F_rec hi | ref16 + agg3d + 0], the window-constant share (g_pw) as the
        // residual, so that the partial sum g_p2 makes no round trip through HBM (256 B per pixel and recursion).  Built, parity-green,
        // measured 0.98 ms against 0.44 + 0.51 ms (same box): the kernel's epilogue (64 KiB of residual loads + 64 KiB of stores issued by
        // the MFMA waves themselves) costs 15 000 of an item's 33 000 cycles (profiles/r06_notes.md section 6, the phase stamps).
        static const bool df2_fuse = getenv("DEMFI_DF2_FUSE") && atoi(getenv("DEMFI_DF2_FUSE")) != 0 &&
                                     !(getenv("DEMFI_WS2") && atoi(getenv("DEMFI_WS2")) == 0);
        if (c->dtype == DEMFI_F16 && df2_fuse) {
            std::vector<int32_t> sel = range(9, 73);                 // F_rec -> sub-layer inputs 0..63
            for (int i = 0; i < 6; ++i) sel.push_back(i);            // S0p, S1p -> 64..69
            for (int i = 73; i < 78; ++i) sel.push_back(i);          // occ_0 -> 70; rflow_t0, rflow_t1 -> 71..74
            sel.insert(sel.end(), a3d_sel.begin(), a3d_sel.end());   // the 8 planes of the recursion -> 75..82
            w_df2 = sub_weight("Dec_first_2", sel, true);
            // ref16 = [S0p, S1p, Stp | rflow_t0, rflow_t1, occ logit | occ_0 | 0]
            df2_m16 = {64, 65, 66, 67, 68, 69, -1, -1, -1, 71, 72, 73, 74, -1, 70, -1};
            df2_m8 = range(75, 83);
        } else
        if (c->dtype == DEMFI_F16) {
            // per recursion: ONE narrow launch over [ref16 (its 11 planes Agg3 holds: t-dependent, recursion-invariant) | agg3d (8 planes of this
            // recursion)] + bias + the window-constant share (g_pw, trunk) -> g_p2; then the F_rec part on the 64 -> 64 kernel
            std::vector<int32_t> sel = range(0, 6);                  // S0p, S1p | occ_0 | rflow_t0, rflow_t1 (agg16 order)
            for (int i = 73; i < 78; ++i) sel.push_back(i);
            sel.insert(sel.end(), a3d_sel.begin(), a3d_sel.end());
            w_dyn = sub_weight("Dec_first_2", sel, true);
            w_rec = sub_weight("Dec_first_2", range(9, 73), false);
            // ref16 channel -> input channel of the sub-layer (sel order): S0p, S1p -> 0..5; Stp unused; rflow_t0, rflow_t1 -> 7..10; the
            // occlusion logit unused; occ_0 (channel 14) -> 6
            dyn_m16 = {0, 1, 2, 3, 4, 5, -1, -1, -1, 7, 8, 9, 10, -1, 6, -1};
        }
        // ============================ recursive boosting, one list per iteration ====================================
        // SepConvGRU (838-857): z | r share their input -> one 128-cout conv (fp32 plan, DEMFI_GRU6=0); round 6, fp16: see fuse_gru
        static const bool gru6_env = !(getenv("DEMFI_GRU6") && atoi(getenv("DEMFI_GRU6")) == 0);
        const bool gru6 = gru6_env && c->dtype == DEMFI_F16;
        std::vector<float> zrw[2], zrb[2];
        const Layer zr_shape[2] = {{128, 128, 1, 5}, {128, 128, 5, 1}};
        for (int s = 0; s < 2 && status >= 0 && !dry && !gru6; ++s) {
            const std::string sfx = std::to_string(s + 1);
            for (const char* g : {"z", "r"}) {
                auto iw = c->weights.find(p + "GB.conv" + g + sfx + ".weight"), ib = c->weights.find(p + "GB.conv" + g + sfx + ".bias");
                if (iw == c->weights.end() || ib == c->weights.end()) {
                    status = demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: GRU weights were not loaded");
                    return;
                }
                zrw[s].insert(zrw[s].end(), iw->second.data.begin(), iw->second.data.end());
                zrb[s].insert(zrb[s].end(), ib->second.data.begin(), ib->second.data.end());
            }
        }
        for (int it = 0; it < N; ++it) {
            OpList& sg = tb > 1 ? c->tb_iter_ops[k][it] : c->iter_ops[k][q][it];
            const Tensor& hin = B[it % 2 ? "frec1" : "frec0"];
            const Tensor& hout = B[it % 2 ? "frec0" : "frec1"];
            {
                // delta16 = the 5 flow / occlusion planes of step `it` as one NHWC record (+ 11 zero channels).  fp16 plan: written by
                // the producer's thin epilogue (dec3#f for step 0, flow_occ.conv2 of the previous recursion otherwise: demfi_conv.pack);
                // fp32 plan (general kernel): a plane-packing launch
                if (c->dtype != DEMFI_F16) {
                    std::vector<const float*> pl;
                    for (int i = 0; i < 5; ++i) pl.push_back(delta_p(it, i));
                    pack(sg, pl, B["delta16"]);
                }
                // the first 8 channels of the record as the piece (5 planes + 3 unused), the other 8 of the k-step as zero padding: the shape
                // the 7x7 kernel's paired-tap mode takes (conv_narrow.hip, P7); the fp32 plan (general kernel) is indifferent
                std::vector<int32_t> m = range(0, 5);
                m.insert(m.end(), 3, -1);
                conv(sg, p + "Mixer.conv_delta1", {fsrc_map(B["delta16"], m)}, {D(fview(B["de1"]), range(0, 32), R)}, H, W);
            }
            conv(sg, p + "Mixer.conv_delta2", {fsrc(B["de1"], 0)}, {D(fview(B["rd64"], 32), range(0, 32), R)}, H, W);
            conv(sg, p + "Mixer.conv_blend1", {fsrc(B["rd64"], 0)}, {D(fview(B["bl1"]), range(0, 32), R)}, H, W);
            conv(sg, p + "Mixer.conv_blend2", {fsrc(B["bl1"], 0)}, {D(fview(B["xb"]), range(0, 64), R)}, H, W);
            const Tensor* h = &hin;
            for (int s = 0; s < 2; ++s) {
                const Tensor& hnext = s == 0 ? B["h1"] : hout;
                const std::string sfx = std::to_string(s + 1);
                if (gru6) {
                    // round 6: r*h, then z + q + blend in one launch (fuse_gru); the three plain layers of the reference module
                    conv(sg, p + "GB.convr" + sfx, {fsrc(*h, 0), fsrc(B["xb"], 64)},
                         {D(fview(B["rh"]), range(0, 64), DEMFI_ACT_NONE, DEMFI_MODE_MUL, fview(*h))}, H, W);
                    conv(sg, p + "GB.convz" + sfx, {fsrc(*h, 0), fsrc(B["xb"], 64)}, {D(fview(B["zb"]), range(0, 64), DEMFI_ACT_SIGMOID)}, H, W);
                } else
                conv(sg, p + "GB.convzr" + sfx, {fsrc(*h, 0), fsrc(B["xb"], 64)},
                     {D(fview(B["zb"]), range(0, 64), DEMFI_ACT_SIGMOID),
                      D(fview(B["rh"]), range(64, 128), DEMFI_ACT_NONE, DEMFI_MODE_MUL, fview(*h))}, H, W, 1, 1, &zrw[s], &zrb[s], &zr_shape[s]);
                conv(sg, p + "GB.convq" + sfx, {fsrc(B["rh"], 0), fsrc(B["xb"], 64)},
                     {D(fview(hnext), range(0, 64), DEMFI_ACT_NONE, DEMFI_MODE_GRU, fview(*h), fview(B["zb"]))}, H, W);
                if (gru6) fuse_gru(sg, p + "GB.step" + sfx);
                h = &hnext;
            }
            conv(sg, p + "flow_occ.conv1", {fsrc(hout, 0)}, {D(fview(B["fo1"]), range(0, 32), R)}, H, W);
            if (c->dtype == DEMFI_F16) pack_next(fview(B["delta16"]), 0);      // step it+1's record for the next recursion's conv_delta1
            conv(sg, p + "flow_occ.conv2", {fsrc(B["fo1"], 0)},
                 {D(delta_v(it + 1, 0), range(0, 5), DEMFI_ACT_NONE, DEMFI_MODE_STORE, delta_v(it, 0))}, H, W);
            // PWB of the recursion; the kernel also writes Agg3's per-recursion planes [St_new | rflow_t0, rflow_t1 | occ]
            // (DeMFInet.py:151-155) as the NHWC record Dec_first_2 reads (agg3d): no plane-packing launch
            warp(sg, "warp_thin", 3, tview(B["sharp1"], 0), tview(B["sharp1"], 3), tview(B["stnew"]), delta_p(it + 1, 0), delta_p(it + 1, 2),
                 delta_p(it + 1, 4), plane(B["occ"], it + 1), tp, ptr(B["agg3d"]));
            if (c->dtype == DEMFI_F16 && df2_fuse) {
                conv(sg, "Dec_first_2#t", {fsrc(hout, 0), fsrc_map(B["ref16"], df2_m16), fsrc_map(B["agg3d"], df2_m8)},
                     {D(fview(B["g_a"]), range(0, 64), R, DEMFI_MODE_STORE, fview(TB["g_pw"]))}, H, W, 1, 1, &w_df2.w, &w_df2.b, &w_df2.shape);
            } else
            if (c->dtype == DEMFI_F16) {
                conv(sg, "Dec_first_2#dyn", {fsrc_map(B["ref16"], dyn_m16), fsrc_map(B["agg3d"], range(11, 19))},
                     {D(fview(B["g_p2"]), range(0, 64), DEMFI_ACT_NONE, DEMFI_MODE_STORE, fview(TB["g_pw"]))}, H, W, 1, 1, &w_dyn.w, &w_dyn.b,
                     &w_dyn.shape);
                conv(sg, "Dec_first_2#rec", {fsrc(hout, 0)}, {D(fview(B["g_a"]), range(0, 64), R, DEMFI_MODE_STORE, fview(B["g_p2"]))}, H, W,
                     1, 1, &w_rec.w, &w_rec.b, &w_rec.shape);
            } else
            conv(sg, "Dec_first_2", {fsrc(hout, 9), fsrc_map(B["agg3s"], agg3s_cin), fsrc_map(B["agg3d"], a3d_sel)},
                 {D(fview(B["g_a"]), range(0, 64), R)}, H, W);
            const Tensor* g = resblocks(sg, "Decoder_res_2", c->hp.num_resb_dec, B["g_a"], B["g_t"], B["g_b"], H, W, 1);
            conv(sg, "Dec_last1_2", {fsrc(*g, 0)}, {D(fview(B["g_t"]), range(0, 64), R)}, H, W);
            sink_for_next = (const demfi_u8_sink*)ptr(B["sink"]);
            sink_iter = it;
            conv(sg, "Dec_last2_2", {fsrc(B["g_t"], 0)},
                 {D(tview(B["finals"], 9 * it + 0), range(0, 3), DEMFI_ACT_NONE, DEMFI_MODE_STORE, tview(B["sharp1"], 0)),
                  D(tview(B["finals"], 9 * it + 3), range(3, 6), DEMFI_ACT_NONE, DEMFI_MODE_STORE, tview(B["sharp1"], 3)),
                  D(tview(B["finals"], 9 * it + 6), range(6, 9), DEMFI_ACT_NONE, DEMFI_MODE_STORE, tview(B["stnew"]))}, H, W);
        }
    }
};

int run_builder(demfi_ctx* c, bool dry)
{
    c->blob_fill = 0;
    c->descs.clear();
    c->pack_cache.clear();
    c->tr_ops.assign(c->n_trunk, OpList());
    c->head_ops.assign(c->n_trunk, std::vector<OpList>(c->n_ctx));
    c->iter_ops.assign(c->n_trunk, std::vector<std::vector<OpList>>(c->n_ctx, std::vector<OpList>(c->N)));
    c->tb_head_ops.assign(c->n_trunk, OpList());
    c->tb_iter_ops.assign(c->n_trunk, std::vector<OpList>(c->N));
    c->fused_now.clear();
    Builder b{c, esz_of(c), c->dtype == DEMFI_F32, dry};
    // the sizing pass records which launches it fused; the bind pass must fuse the same ones (their untouched scratch has no memory)
    auto done = [&]() {
        if (b.status < 0) return b.status;
        if (dry) c->fused_dry = c->fused_now;
        else if (c->fused_now != c->fused_dry)
            return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: the bound plan fuses %d launches, the sizing pass fused %d (or others): the "
                                   "arena gave their scratch buffers no memory -- set DEMFI_ARENA=0 and report", (int)c->fused_now.size(), (int)c->fused_dry.size());
        return b.status;
    };
    if (c->op_kind) {
        b.build_operator();
        return done();
    }
    for (int k = 0; k < c->n_trunk && b.status >= 0; ++k) {
        b.build_trunk(k);
        for (int q = 0; q < c->n_ctx && b.status >= 0; ++q) b.build_t(k, q);
        if (c->n_ctx > 1 && b.status >= 0) {                    // the batched plan over all contexts of this trunk set
            b.tb = c->n_ctx; b.tb_k = k;
            b.build_t(k, 0);
            b.tb = 1;
        }
    }
    return done();
}

// ---- workspace arena (round 5) -----------------------------------------------------------------------------------
// Every pointer an op reads / writes (write = true).  The scratch buffer between the two convolutions of a fused residual block is
// not touched by the fused launch.
void op_accesses(const demfi_ctx* c, const demfi_op& op, std::vector<std::pair<const void*, bool>>& out)
{
    auto rd = [&](const void* p) { if (p) out.push_back({p, false}); };
    auto wr = [&](const void* p) { if (p) out.push_back({p, true}); };
    auto conv_in = [&](const demfi_conv& d) { for (int i = 0; i < d.n_pieces; ++i) rd(d.pieces[i].v.ptr); };
    auto conv_out = [&](const demfi_conv& d) {
        for (int i = 0; i < d.n_segs; ++i) { rd(d.segs[i].res.ptr); rd(d.segs[i].aux.ptr); wr(d.segs[i].dst.ptr); }
        wr(d.pack.ptr);
    };
    switch (op.kind) {
    case DEMFI_OP_CONV: conv_in(c->descs[op.conv]); conv_out(c->descs[op.conv]); break;
    case DEMFI_OP_RESBLOCK: conv_in(c->descs[op.conv]); conv_out(c->descs[op.nch]); break;
    case DEMFI_OP_GRU_R: conv_in(c->descs[op.conv]); conv_out(c->descs[op.conv]); break;
    case DEMFI_OP_GRU_ZQ: {                                      // reads h, x, r*h; writes h'; the z buffer (convz's dst == convq's aux) is not touched
        const demfi_conv& dq = c->descs[op.nch];
        conv_in(c->descs[op.conv]); conv_in(dq);
        for (int i = 0; i < dq.n_segs; ++i) { rd(dq.segs[i].res.ptr); wr(dq.segs[i].dst.ptr); }
        break;
    }
    case DEMFI_OP_PACK: for (int i = 0; i < 32; ++i) rd(op.p[i]); wr(op.o.ptr); break;
    case DEMFI_OP_VIZ: rd(op.a.ptr); rd(op.b.ptr); rd(op.p[1]); if (op.conv == 1) { rd(op.p[0]); wr(op.p[1]); } wr(op.p[0]); break;
    case DEMFI_OP_S2D: case DEMFI_OP_OVERLAY: rd(op.p[0]); wr(op.p[1]); break;
    case DEMFI_OP_FGAC: rd(op.a.ptr); rd(op.p[0]); wr(op.o.ptr); break;
    case DEMFI_OP_FGAC_WINDOW: rd(op.a.ptr); rd(op.b.ptr); rd(op.p[0]); wr(op.o.ptr); break;
    case DEMFI_OP_AVG_POOL: rd(op.a.ptr); wr(op.o.ptr); break;
    case DEMFI_OP_GATE: rd(op.p[0]); rd(op.a.ptr); rd(op.b.ptr); wr(op.o.ptr); break;
    case DEMFI_OP_CFR: rd(op.p[0]); rd(op.p[1]); rd(op.p[2]); rd(op.p[4]); wr(op.p[2]); wr(op.p[3]); wr(op.p[5]); rd(op.t); break;
    case DEMFI_OP_WARP: rd(op.a.ptr); rd(op.b.ptr); rd(op.p[0]); rd(op.p[1]); rd(op.p[2]); rd(op.t); wr(op.o.ptr); wr(op.p[3]); wr(op.p[4]); break;
    default: break;
    }
}

// Liveness plan of one buffer set from its launch sequence (built on the UNALIASED layout of the sizing pass, where an address
// names one buffer).  Candidates: the buffers in `allow` whose first access is a write by an op of `seq` and which no op of
// `foreign` (another segment) touches; a candidate lives from its first to its last access (buffers that carry state from one
// recursion to the next are accessed in several: their interval spans them).  Everything else keeps memory of its own: inputs,
// outputs the host reads, buffers that rely on the zero-filled workspace (the CFR accumulator, zero-padded records).  Footprints
// (all n_ctx copies of a buffer: the tensor-major layout stays) are packed first-fit, largest first.
// does op (re)write EVERY element of tensor t (all images, all channels)?  Then whatever t held before is dead: its live range
// may end at the previous access and a new one starts here (per-recursion scratch is alive only inside each recursion).
bool op_overwrites(const demfi_ctx* c, const demfi_op& op, const Tensor& t, int64_t t_addr)
{
    if (t.kind != 0) return false;
    auto conv_full = [&](const demfi_conv& d) {
        if (d.H != t.d[1] || d.W != t.d[2] || d.batch != t.d[0]) return false;
        for (int sg = 0; sg < d.n_segs; ++sg) {
            const demfi_seg& g = d.segs[sg];
            if ((int64_t)(intptr_t)g.dst.ptr != t_addr || g.mode != DEMFI_MODE_STORE && g.mode != DEMFI_MODE_MUL && g.mode != DEMFI_MODE_GRU) continue;
            if (g.scale != 1 || g.dst.sc != 1 || g.dst.sx != t.d[3]) continue;
            int n = 0;
            for (int o = 0; o < d.cout_pad / 8; ++o) if (d.oct_seg[o] == sg) n += d.oct_n[o];
            if (n == t.d[3]) return true;
        }
        return false;
    };
    if (op.kind == DEMFI_OP_CONV) return conv_full(c->descs[op.conv]);
    if (op.kind == DEMFI_OP_RESBLOCK || op.kind == DEMFI_OP_GRU_ZQ) return conv_full(c->descs[op.nch]);
    if (op.kind == DEMFI_OP_GRU_R) return conv_full(c->descs[op.conv]);
    if (op.kind == DEMFI_OP_PACK) return (int64_t)(intptr_t)op.o.ptr == t_addr && op.nch == t.d[3] && t.d[0] == 1;
    return false;
}

ArenaPlan plan_arena(const demfi_ctx* c, const BufSet& set, int rep, const std::vector<const OpList*>& seq,
                     const std::vector<const OpList*>& foreign, const std::vector<std::string>& allow)
{
    struct Iv { std::string name; int64_t size; std::vector<std::pair<int, int>> live; bool ok = true; int64_t off = -1; };
    std::vector<Iv> iv;
    const char* only = getenv("DEMFI_ARENA_ONLY");              // debugging: restrict the arena to the named buffers ("a,b,c")
    for (const auto& n : allow) {
        if (only && (std::string(",") + only + ",").find("," + n + ",") == std::string::npos) continue;
        auto it = set.find(n);
        if (it != set.end()) iv.push_back({n, (Layout::footprint(it->second, rep) + 255) & ~255ll});
    }
    auto find = [&](const void* p) -> Iv* {
        const int64_t a = (int64_t)(intptr_t)p;                  // sizing pass: base == nullptr, pointers are offsets
        for (auto& x : iv) {
            const Tensor& t = set.find(x.name)->second;
            if (a >= t.off && a < t.off + t.bytes) return &x;
        }
        return nullptr;
    };
    std::vector<std::pair<const void*, bool>> acc;
    int idx = 0;
    for (const OpList* ops : seq)
        for (const demfi_op& op : *ops) {
            acc.clear();
            op_accesses(c, op, acc);
            for (int pass = 0; pass < 2; ++pass)                 // an op's reads come before its writes
                for (auto& a : acc) {
                    if ((int)a.second != pass) continue;
                    Iv* x = find(a.first);
                    if (!x) continue;
                    const Tensor& t = set.find(x->name)->second;
                    if (x->live.empty()) {
                        if (!a.second) x->ok = false;             // read before any write: it relies on what the workspace held
                        x->live.push_back({idx, idx});
                    } else if (a.second && x->live.back().second < idx && op_overwrites(c, op, t, t.off)) {
                        x->live.push_back({idx, idx});            // everything it held is replaced: a new live range
                    } else x->live.back().second = idx;
                }
            ++idx;
        }
    for (const OpList* ops : foreign)
        for (const demfi_op& op : *ops) {
            acc.clear();
            op_accesses(c, op, acc);
            for (auto& a : acc) if (Iv* x = find(a.first)) x->ok = false;
        }
    ArenaPlan pl;
    std::vector<Iv*> todo;
    for (auto& x : iv) {
        if (x.live.empty()) { pl.off[x.name] = 0; continue; }    // never touched (the scratch of fused residual blocks): no memory
        if (x.ok) todo.push_back(&x);
    }
    std::sort(todo.begin(), todo.end(), [](const Iv* a, const Iv* b) { return a->size != b->size ? a->size > b->size : a->live[0].first < b->live[0].first; });
    auto together = [](const Iv* a, const Iv* b) {
        for (auto& p : a->live) for (auto& q : b->live) if (!(p.second < q.first || q.second < p.first)) return true;
        return false;
    };
    std::vector<Iv*> placed;
    if (rep > 0) {
        // Per-t set: the arena is a row of SLOTS.  A slot holds, per context, S bytes (S = the largest member: the 3-image buffers
        // of D1); context q's share of slot j is [j * rep * S + q * S, + S).  A member lives at (slot, offset < S) with context stride
        // S, so everything context q ever touches lies inside ITS shares: per-t contexts stay independent of one another (they
        // may run concurrently on different streams, demfi_forward_t) while buffers of one context that are never alive together
        // share memory.  A multi-image member fills a slot exactly (its images tile the context stride, as the batched plan needs).
        int64_t S = 0;
        for (Iv* x : todo) x->size = (Layout::footprint(set.find(x->name)->second, 1) + 255) & ~255ll;     // bytes per context
        for (Iv* x : todo) S = std::max(S, x->size);
        std::vector<Iv*> keep;
        for (Iv* x : todo) {
            const Tensor& t = set.find(x->name)->second;
            if (t.kind == 0 && t.d[0] > 1 && ((t.bytes + 15) & ~15ll) != S) { x->ok = false; continue; }   // its images would not tile the slot stride
            keep.push_back(x);
        }
        todo.swap(keep);
        std::sort(todo.begin(), todo.end(), [](const Iv* a, const Iv* b) { return a->size != b->size ? a->size > b->size : a->live[0].first < b->live[0].first; });
        std::vector<int> slot_of;
        int n_slots = 0;
        for (Iv* x : todo) {
            int64_t o = -1;
            int sj = 0;
            for (; o < 0; ++sj) {
                std::vector<std::pair<int64_t, int64_t>> busy;
                for (size_t i = 0; i < placed.size(); ++i)
                    if (slot_of[i] == sj && together(x, placed[i])) busy.push_back({placed[i]->off, placed[i]->off + placed[i]->size});
                std::sort(busy.begin(), busy.end());
                int64_t f = 0;
                for (auto& b : busy) { if (f + x->size <= b.first) break; f = std::max(f, b.second); }
                if (f + x->size <= S) { o = f; break; }
            }
            x->off = o;
            placed.push_back(x);
            slot_of.push_back(sj);
            n_slots = std::max(n_slots, sj + 1);
            pl.off[x->name] = (int64_t)sj * rep * S + o;
        }
        pl.cstride = S;
        pl.size = (int64_t)n_slots * rep * S;
        for (size_t i = 0; i < placed.size(); ++i) placed[i]->off += (int64_t)slot_of[i] * rep * S;      // for the debug print
    } else
    for (Iv* x : todo) {
        std::vector<std::pair<int64_t, int64_t>> busy;           // address ranges of placed buffers alive at the same time
        for (Iv* y : placed) if (together(x, y)) busy.push_back({y->off, y->off + y->size});
        std::sort(busy.begin(), busy.end());
        int64_t o = 0;
        for (auto& b : busy) { if (o + x->size <= b.first) break; o = std::max(o, b.second); }
        x->off = o;
        placed.push_back(x);
        pl.off[x->name] = o;
        pl.size = std::max(pl.size, o + x->size);
    }
    if (getenv("DEMFI_ARENA_DEBUG"))
        for (auto& x : iv) {
            fprintf(stderr, "   %-8s ok=%d off=%8.1f MB size=%7.1f MB live", x.name.c_str(), (int)x.ok, x.off / 1e6, x.size / 1e6);
            for (auto& p : x.live) fprintf(stderr, " [%d,%d]", p.first, p.second);
            fprintf(stderr, "\n");
        }
    pl.size = std::max<int64_t>(pl.size, 256);                   // never-touched members point at the arena's first bytes
    pl.on = !placed.empty();
    if (!pl.on) pl.off.clear();
    return pl;
}

int run_ops(demfi_ctx* c, const OpList& ops, void* stream)
{
    for (const demfi_op& op : ops) {
        const int st = demfi_run_op(c, &op, stream);
        if (st < 0) return st;
    }
    return DEMFI_OK;
}

bool check_idx(const demfi_ctx* c, int trunk, int q)
{
    return c && trunk >= 0 && trunk < c->n_trunk && q >= 0 && q < c->n_ctx;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int demfi_conv_build(int dtype, int H, int W, int stride, int batch, const float* w, const float* bias, int cout, int cin,
                                int kh, int kw, const demfi_conv_src* srcs, int n_srcs, const demfi_conv_dst* dsts, int n_dsts,
                                demfi_conv* desc, void* wpack, int64_t* wpack_bytes, float* bias_packed, int32_t* cout_pad)
{
    if (!w || !desc || !wpack_bytes || !cout_pad || H <= 0 || W <= 0 || batch <= 0 || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_conv_build: bad arguments");
    BuiltConv bc;
    const bool size_only = wpack == nullptr;
    int st = build_conv(dtype, H, W, stride, batch, w, bias, cout, cin, kh, kw, srcs, n_srcs, dsts, n_dsts, bc, size_only, "demfi_conv_build");
    if (st < 0) return st;
    *cout_pad = bc.d.cout_pad;
    if (size_only) {
        *wpack_bytes = bc.wbytes;
        *desc = bc.d;
        return DEMFI_OK;
    }
    *wpack_bytes = (int64_t)bc.wpack.size();
    memcpy(wpack, bc.wpack.data(), bc.wpack.size());
    if (bias_packed) memcpy(bias_packed, bc.bias.data(), bc.bias.size() * 4);
    *desc = bc.d;
    return DEMFI_OK;
}

extern "C" int demfi_ctx_create(int H, int W, int max_updates, int dtype, const demfi_hparams* hp, int n_trunk, int n_ctx, demfi_ctx** out)
{
    if (!out) return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_create: null out");
    if (H <= 0 || W <= 0 || H % 8 || W % 8)
        return demfi_set_error(DEMFI_ERR_ARG, "DeMFI-Net needs H, W multiples of 8 (the harness pads to 32): got %dx%d", H, W);
    if (max_updates < 1 || max_updates > 64 || (dtype != DEMFI_F16 && dtype != DEMFI_F32) || n_trunk < 1 || n_ctx < 1 || n_trunk > 8 || n_ctx > 16)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_create: max_updates / dtype / context counts");
    demfi_hparams h = {64, 2, 5, 5, 1, 0, 0, 0};
    if (hp) h = *hp;
    if (h.nf != 64 || h.scale_factor != 2)
        return demfi_set_error(DEMFI_ERR_ARG, "the HIP path is built for nf=64, scale_factor=2 (the released configuration)");
    if (h.num_resb_facfb < 0 || h.num_resb_dec < 0 || h.num_resb_facfb > 32 || h.num_resb_dec > 32)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_create: residual block counts");
    // fgac_rr / fgac_sr: the radii hard-coded to 0 at DeMFInet.py:401-402; > 0 selects the generalised window FGAC
    // (demfi_fgac_window, both path dtypes; flags bit 0 = index map: 0 reference code, 1 pixel-centred window)
    if (h.fgac_rr < 0 || h.fgac_rr > 2 || h.fgac_sr < 0 || h.fgac_sr > 4 || (h.flags & ~(DEMFI_HP_FGAC_CENTRED | DEMFI_HP_EXTRAS)))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_create: fgac_rr in 0..2, fgac_sr in 0..4, map in {0,1}");
    if (h.fgac_rr == 0 && h.fgac_sr != 0)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_create: fgac_sr > 0 needs fgac_rr > 0 (the pooled point-wise form is not built)");
    demfi_ctx* c = new demfi_ctx();
    c->H = H; c->W = W; c->N = max_updates; c->dtype = dtype; c->n_trunk = n_trunk; c->n_ctx = n_ctx; c->hp = h;
    layer_table(c);
    // sizing pass: lay the buffers out, walk the plan without weights to learn the exact size of the packed blob and the
    // number of descriptors, then lay everything out for real
    compute_layout(c, 0, 0);
    int st = run_builder(c, true);
    if (st < 0) { delete c; return st; }
    // the workspace arena: buffers of a set that are never alive together share memory (DEMFI_ARENA=0: one region per buffer,
    // the layout of rounds 1-4; results are bit-identical either way)
    static const bool arena_on = !(getenv("DEMFI_ARENA") && atoi(getenv("DEMFI_ARENA")) == 0);
    if (arena_on && !c->op_kind) {
        std::vector<const OpList*> per_t = {&c->head_ops[0][0]}, tr = {&c->tr_ops[0]}, none;
        for (int it = 0; it < c->N; ++it) per_t.push_back(&c->iter_ops[0][0][it]);
        std::vector<const OpList*> per_t_all = per_t;
        if (c->n_ctx > 1) { per_t_all.push_back(&c->tb_head_ops[0]); for (int it = 0; it < c->N; ++it) per_t_all.push_back(&c->tb_iter_ops[0][it]); }
        const ArenaPlan pt = plan_arena(c, c->t_bufs[0][0], c->n_ctx, per_t, none,
                                        {"Ft", "u1", "u2", "u3", "d0", "d1", "d2", "rF", "dec_a", "dec_t", "dec_b", "frec0", "frec1", "re1", "rd64", "de1",
                                         "bl1", "xb", "zb", "rh", "h1", "fo1", "g_a", "g_t", "g_b", "g_p2", "misc16", "ref16", "ref32", "agg3s"});
        const ArenaPlan ptr_ = plan_arena(c, c->tr_bufs[0], 0, tr, per_t_all,
                                          {"s2d", "f1", "x0", "grow", "gffcat", "g0", "g1", "up", "enc_a", "enc_t", "enc_b", "rk", "skk", "rkp", "skp",
                                           "smp", "E", "wg"});
        c->arena_t = pt;
        c->arena_tr = ptr_;
        if (getenv("DEMFI_ARENA_DEBUG")) {
            for (auto* pl : {&c->arena_t, &c->arena_tr}) {
                fprintf(stderr, "arena %s: %.1f MB\n", pl == &c->arena_t ? "per-t (all contexts)" : "trunk", pl->size / 1e6);
                for (auto& kv : pl->off) fprintf(stderr, "   %-8s at %.1f MB\n", kv.first.c_str(), kv.second / 1e6);
            }
        }
    }
    compute_layout(c, c->blob_fill, (int64_t)c->descs.size());
    c->descs.clear();
    *out = c;
    return DEMFI_OK;
}

// ---- single-call operator contexts (ABI v7; SURVEY.md 8b names demfi_gru_sep / demfi_fgac) --------------------------------------
static int operator_create(int kind, int batch, int H, int W, int dtype, demfi_ctx** out)
{
    if (!out) return demfi_set_error(DEMFI_ERR_ARG, "operator context: null out");
    if (batch < 1 || batch > 64 || H < 8 || W < 8 || (dtype != DEMFI_F16 && dtype != DEMFI_F32))
        return demfi_set_error(DEMFI_ERR_ARG, "operator context: batch 1..64, H, W >= 8, dtype F16 / F32");
    demfi_ctx* c = new demfi_ctx();
    c->H = H; c->W = W; c->N = 1; c->dtype = dtype; c->n_trunk = 1; c->n_ctx = 1; c->op_kind = kind; c->op_batch = batch;
    c->hp = {64, 2, 0, 0, 1, 0, 0, 0};
    layer_table(c);
    compute_layout(c, 0, 0);
    int st = run_builder(c, true);
    if (st < 0) { delete c; return st; }
    compute_layout(c, c->blob_fill, (int64_t)c->descs.size());
    c->descs.clear();
    *out = c;
    return DEMFI_OK;
}

extern "C" int demfi_gru_sep_create(int batch, int H, int W, int dtype, demfi_ctx** out) { return operator_create(1, batch, H, W, dtype, out); }
extern "C" int demfi_fgac_create(int batch, int H, int W, int dtype, demfi_ctx** out) { return operator_create(2, batch, H, W, dtype, out); }

extern "C" int demfi_operator_run(demfi_ctx* c, void* stream)
{
    if (!c || !c->op_kind || !c->bound || c->on_host)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_operator_run: not an operator context bound to device memory");
    return run_ops(c, c->tr_ops[0], stream);
}

extern "C" int demfi_ctx_destroy(demfi_ctx* c)
{
    delete c;
    return DEMFI_OK;
}

extern "C" int demfi_load_weight(demfi_ctx* c, const char* name, const float* host, const int64_t* shape, int ndim)
{
    if (!c || !name || !host || !shape || ndim < 1 || ndim > 5) return demfi_set_error(DEMFI_ERR_ARG, "demfi_load_weight: bad arguments");
    if (c->bound) return demfi_set_error(DEMFI_ERR_ARG, "demfi_load_weight: context already bound (create a new one to change weights)");
    std::string n(name);
    const bool is_w = n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0;
    const bool is_b = n.size() > 5 && n.compare(n.size() - 5, 5, ".bias") == 0;
    if (!is_w && !is_b) return demfi_set_error(DEMFI_ERR_ARG, "demfi_load_weight: '%s' is neither a .weight nor a .bias key", name);
    const std::string layer = n.substr(0, n.size() - (is_w ? 7 : 5));
    auto it = c->table.find(layer);
    if (it == c->table.end()) return demfi_set_error(DEMFI_ERR_ARG, "demfi_load_weight: unknown state_dict key '%s'", name);
    const Layer& l = it->second;
    int64_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    bool ok;
    if (is_b) ok = ndim == 1 && shape[0] == l.cout;
    else if (ndim == 4) ok = shape[0] == l.cout && shape[1] == l.cin && shape[2] == l.kh && shape[3] == l.kw;
    else ok = ndim == 5 && shape[0] == l.cout && shape[1] == l.cin && shape[2] == 1 && shape[3] == l.kh && shape[4] == l.kw;   // Conv3d (1,k,k)
    if (!ok) return demfi_set_error(DEMFI_ERR_ARG, "demfi_load_weight: shape of '%s' does not match [%d,%d,%d,%d]", name, l.cout, l.cin, l.kh, l.kw);
    Weight& w = c->weights[n];
    w.data.assign(host, host + numel);
    w.shape.assign(shape, shape + ndim);
    return DEMFI_OK;
}

extern "C" int64_t demfi_ctx_workspace_bytes(const demfi_ctx* c) { return c ? c->total : 0; }

extern "C" int64_t demfi_workspace_bytes(int H, int W, int max_updates, int dtype, int n_trunk, int n_ctx)
{
    demfi_ctx* c = nullptr;
    if (demfi_ctx_create(H, W, max_updates, dtype, nullptr, n_trunk, n_ctx, &c) < 0) return -1;
    const int64_t n = c->total;
    delete c;
    return n;
}

extern "C" int demfi_ctx_bind(demfi_ctx* c, void* workspace, int64_t bytes, int on_host, void* stream)
{
    if (!c || !workspace) return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: null argument");
    if (c->bound) return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: already bound");
    if (bytes < c->total) return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: workspace of %lld B < %lld B", (long long)bytes, (long long)c->total);
    if (((uintptr_t)workspace) & 255) return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: workspace must be 256-byte aligned");
    c->base = (char*)workspace;
    c->on_host = on_host != 0;
    c->host_blob.assign(c->w_bytes, 0);
    const int st = run_builder(c, false);
    if (st < 0) return st;
    const int64_t desc_bytes = (int64_t)c->descs.size() * (int64_t)sizeof(demfi_conv);
    if ((int64_t)c->descs.size() != c->n_descs || c->blob_fill > c->w_bytes)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_bind: plan differs from the sizing pass (%d descriptors, %lld B of weights)",
                               (int)c->descs.size(), (long long)c->blob_fill);
    if (c->on_host) {
        memcpy(c->base + c->w_region, c->host_blob.data(), c->w_bytes);
        memcpy(c->base + c->desc_off, c->descs.data(), desc_bytes);
    } else {
        hipStream_t st = (hipStream_t)stream;
        DEMFI_HIP_CHECK(hipMemcpyAsync(c->base + c->w_region, c->host_blob.data(), c->w_bytes, hipMemcpyHostToDevice, st));
        DEMFI_HIP_CHECK(hipMemcpyAsync(c->base + c->desc_off, c->descs.data(), desc_bytes, hipMemcpyHostToDevice, st));
        DEMFI_HIP_CHECK(hipStreamSynchronize(st));
    }
    c->host_blob.clear();
    c->host_blob.shrink_to_fit();
    c->weights.clear();                                          // the fp32 copies are not needed once packed
    c->bound = true;
    return DEMFI_OK;
}

extern "C" int demfi_ctx_weight_region(const demfi_ctx* c, int64_t* offset, int64_t* bytes)
{
    if (!c || !offset || !bytes) return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_weight_region: null argument");
    *offset = c->w_region;
    *bytes = c->w_bytes;
    return DEMFI_OK;
}

extern "C" int demfi_ctx_buffer(const demfi_ctx* c, int trunk, int q, const char* name, int64_t* offset, int32_t* kind, int32_t dims[4])
{
    if (!c || !name || !offset || trunk < 0 || trunk >= c->n_trunk || q < -1 || q >= c->n_ctx)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_buffer: bad arguments");
    const BufSet& s = q < 0 ? c->tr_bufs[trunk] : c->t_bufs[trunk][q];
    auto it = s.find(name);
    if (it == s.end()) return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_buffer: no buffer '%s' in %s context", name, q < 0 ? "the trunk" : "the per-t");
    *offset = it->second.off;
    if (kind) *kind = it->second.kind;
    if (dims) for (int i = 0; i < 4; ++i) dims[i] = it->second.d[i];
    return DEMFI_OK;
}

extern "C" int demfi_ingest_u8(demfi_ctx* c, int trunk, const uint8_t* const* frames, int h, int w, void* stream)
{
    if (!c || !c->bound || c->on_host || trunk < 0 || trunk >= c->n_trunk)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_ingest_u8: context not bound to device memory / bad trunk index");
    BufSet& B = c->tr_bufs[trunk];
    return demfi_u8_ingest(frames, h, w, (float*)(c->base + B["x"].off), c->base + B["s2d"].off, (float*)(c->base + B["overlay"].off),
                           c->dtype, c->H, c->W, stream);
}

extern "C" int demfi_forward_trunk_body(demfi_ctx* c, int trunk, void* stream)
{
    if (!c || !c->bound || c->on_host || trunk < 0 || trunk >= c->n_trunk)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_forward_trunk_body: context not bound to device memory / bad trunk index");
    const OpList& ops = c->tr_ops[trunk];                        // ops 0, 1 = s2d, overlay: done by demfi_ingest_u8
    for (size_t i = 2; i < ops.size(); ++i) {
        const int st = demfi_run_op(c, &ops[i], stream);
        if (st < 0) return st;
    }
    return DEMFI_OK;
}

extern "C" int demfi_forward_trunk(demfi_ctx* c, int trunk, const float* x, void* stream)
{
    if (!c || !c->bound || c->on_host || trunk < 0 || trunk >= c->n_trunk)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_forward_trunk: context not bound to device memory / bad trunk index");
    const Tensor& xb = c->tr_bufs[trunk]["x"];
    if (x && (const char*)x != c->base + xb.off)
        DEMFI_HIP_CHECK(hipMemcpyAsync(c->base + xb.off, x, xb.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return run_ops(c, c->tr_ops[trunk], stream);
}

extern "C" int demfi_forward_t(demfi_ctx* c, int trunk, int q, int n_updates, void* stream)
{
    if (!c || !c->bound || c->on_host || !check_idx(c, trunk, q))
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_forward_t: context not bound to device memory / bad context index");
    if (n_updates < 1 || n_updates > c->N)
        return demfi_set_error(DEMFI_ERR_ARG, "num_update=%d outside 1..%d the context was built for", n_updates, c->N);
    int st = run_ops(c, c->head_ops[trunk][q], stream);
    for (int it = 0; it < n_updates && st >= 0; ++it) st = run_ops(c, c->iter_ops[trunk][q][it], stream);
    return st;
}

extern "C" int demfi_forward_tb(demfi_ctx* c, int trunk, int n_updates, void* stream)
{
    if (!c || !c->bound || c->on_host || trunk < 0 || trunk >= c->n_trunk)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_forward_tb: context not bound to device memory / bad trunk index");
    if (c->n_ctx < 2) return demfi_set_error(DEMFI_ERR_ARG, "demfi_forward_tb: the context has one per-t context (use demfi_forward_t)");
    if (n_updates < 1 || n_updates > c->N)
        return demfi_set_error(DEMFI_ERR_ARG, "num_update=%d outside 1..%d the context was built for", n_updates, c->N);
    int st = run_ops(c, c->tb_head_ops[trunk], stream);
    for (int it = 0; it < n_updates && st >= 0; ++it) st = run_ops(c, c->tb_iter_ops[trunk][it], stream);
    return st;
}

// In a recursion the PWB + D2 tail (warp_thin .. Dec_last2_2) only produces that recursion's frames: the state the next
// recursion reads is F_rec and the flow / occlusion logits (DeMFInet.py:130-137; Agg3 and D2, 146-165, feed Sharps_final only).
static size_t d2_start(const OpList& ops)
{
    for (size_t i = 0; i < ops.size(); ++i)
        if (ops[i].kind == DEMFI_OP_WARP && ops[i].nch == 3) return i;
    return ops.size();
}

extern "C" int demfi_forward_tb_final(demfi_ctx* c, int trunk, int n_updates, void* stream)
{
    if (!c || !c->bound || c->on_host || trunk < 0 || trunk >= c->n_trunk)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_forward_tb_final: context not bound to device memory / bad trunk index");
    if (c->n_ctx < 2) return demfi_set_error(DEMFI_ERR_ARG, "demfi_forward_tb_final: the context has one per-t context");
    if (n_updates < 1 || n_updates > c->N)
        return demfi_set_error(DEMFI_ERR_ARG, "num_update=%d outside 1..%d the context was built for", n_updates, c->N);
    int st = run_ops(c, c->tb_head_ops[trunk], stream);
    for (int it = 0; it < n_updates && st >= 0; ++it) {
        const OpList& ops = c->tb_iter_ops[trunk][it];
        const size_t n = it + 1 < n_updates ? d2_start(ops) : ops.size();
        for (size_t i = 0; i < n && st >= 0; ++i) st = demfi_run_op(c, &ops[i], stream);
    }
    return st;
}

static const OpList* seg_ops(const demfi_ctx* c, int segment, int trunk, int q, int iter)
{
    if (!c || !c->bound || trunk < 0 || trunk >= c->n_trunk) return nullptr;
    if (segment == DEMFI_SEG_TRUNK) return &c->tr_ops[trunk];
    if (segment == DEMFI_SEG_TB_HEAD) return &c->tb_head_ops[trunk];
    if (segment == DEMFI_SEG_TB_ITER && iter >= 0 && iter < c->N) return &c->tb_iter_ops[trunk][iter];
    if (q < 0 || q >= c->n_ctx) return nullptr;
    if (segment == DEMFI_SEG_T_HEAD) return &c->head_ops[trunk][q];
    if (segment == DEMFI_SEG_ITER && iter >= 0 && iter < c->N) return &c->iter_ops[trunk][q][iter];
    return nullptr;
}

extern "C" int demfi_ctx_num_ops(const demfi_ctx* c, int segment, int trunk, int q, int iter)
{
    const OpList* o = seg_ops(c, segment, trunk, q, iter);
    if (!o) return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_num_ops: bad segment / context (or context not bound)");
    return (int)o->size();
}

extern "C" int demfi_ctx_get_op(const demfi_ctx* c, int segment, int trunk, int q, int iter, int index, demfi_op* out)
{
    const OpList* o = seg_ops(c, segment, trunk, q, iter);
    if (!o || !out || index < 0 || index >= (int)o->size()) return demfi_set_error(DEMFI_ERR_ARG, "demfi_ctx_get_op: bad arguments");
    *out = (*o)[index];
    return DEMFI_OK;
}

extern "C" int demfi_ctx_num_convs(const demfi_ctx* c) { return c ? (int)c->descs.size() : 0; }

extern "C" const demfi_conv* demfi_ctx_conv_desc(const demfi_ctx* c, int index)
{
    if (!c || index < 0 || index >= (int)c->descs.size()) return nullptr;
    return &c->descs[index];
}

extern "C" int demfi_run_op(demfi_ctx* c, const demfi_op* op, void* stream)
{
    if (!c || !op || !c->bound || c->on_host) return demfi_set_error(DEMFI_ERR_ARG, "demfi_run_op: context not bound to device memory");
    const int H = c->H, W = c->W;
    switch (op->kind) {
    case DEMFI_OP_CONV:
        if (op->conv < 0 || op->conv >= (int)c->descs.size()) return demfi_set_error(DEMFI_ERR_ARG, "demfi_run_op: descriptor index");
        return demfi_conv2d(&c->descs[op->conv], (const demfi_conv*)(c->base + c->desc_off) + op->conv, stream);
    case DEMFI_OP_RESBLOCK:
        if (op->conv < 0 || op->conv >= (int)c->descs.size() || op->nch < 0 || op->nch >= (int)c->descs.size())
            return demfi_set_error(DEMFI_ERR_ARG, "demfi_run_op: descriptor index");
        return demfi_resblock3x3_c64(&c->descs[op->conv], &c->descs[op->nch], stream);
    case DEMFI_OP_GRU_R:
        if (op->conv < 0 || op->conv >= (int)c->descs.size()) return demfi_set_error(DEMFI_ERR_ARG, "demfi_run_op: descriptor index");
        return demfi_gru_r(&c->descs[op->conv], stream);
    case DEMFI_OP_GRU_ZQ:
        if (op->conv < 0 || op->conv >= (int)c->descs.size() || op->nch < 0 || op->nch >= (int)c->descs.size())
            return demfi_set_error(DEMFI_ERR_ARG, "demfi_run_op: descriptor index");
        return demfi_gru_zq(&c->descs[op->conv], &c->descs[op->nch], stream);
    case DEMFI_OP_PACK:
        if (op->bt.nb > 1)
            return demfi_pack_planes_batched((const float* const*)op->p, op->nch, op->o.ptr, c->dtype, op->o.sx, H, W, &op->bt, stream);
        return demfi_pack_planes((const float* const*)op->p, op->nch, op->o.ptr, c->dtype, op->o.sx, H, W, stream);
    case DEMFI_OP_VIZ:
        if (op->conv == 0) return demfi_absmean_map(&op->a, op->b.ptr ? &op->b : nullptr, (float*)op->p[0], op->nch, H, W, stream);
        if (op->conv == 1) return demfi_minmax_normalize((float*)op->p[0], (int64_t)H * W, (float*)op->p[1], stream);
        if (op->conv == 2) return demfi_one_minus((const float*)op->p[1], (float*)op->p[0], (int64_t)H * W, stream);
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_run_op: visualisation sub-op %d", op->conv);
    case DEMFI_OP_S2D:
        return demfi_space_to_depth((const float*)op->p[0], (void*)op->p[1], c->dtype, H, W, stream);
    case DEMFI_OP_OVERLAY:
        return demfi_overlay_mean((const float*)op->p[0], (float*)op->p[1], H, W, stream);
    case DEMFI_OP_FGAC:
        return demfi_fgac_gather(&op->a, (const float*)op->p[0], &op->o, op->nch, H, W, nullptr, stream);
    case DEMFI_OP_FGAC_WINDOW:
        return demfi_fgac_window(&op->a, &op->b, (const float*)op->p[0], &op->o, op->nch, H, W, op->conv, op->_pad, nullptr, stream);
    case DEMFI_OP_AVG_POOL:
        return demfi_avg_pool_fat(&op->a, &op->o, op->nch, H, W, op->conv, stream);
    case DEMFI_OP_GATE:
        return demfi_gate_blend((const float*)op->p[0], &op->a, &op->b, &op->o, op->nch, H, W, stream);
    case DEMFI_OP_CFR:
        if (op->p[5])
            return demfi_cfr_flow_align_pack((const float*)op->p[0], (const float*)op->p[1], (const float*)op->p[4], (const float*)op->t, H, W,
                                             (int64_t*)op->p[2], (float*)op->p[3], (void*)op->p[5], c->dtype, op->bt.nb > 1 ? &op->bt : nullptr, stream);
        if (op->bt.nb > 1)
            return demfi_cfr_flow_align_batched((const float*)op->p[0], (const float*)op->p[1], (const float*)op->t, H, W, (int64_t*)op->p[2],
                                                (float*)op->p[3], &op->bt, stream);
        return demfi_cfr_flow_align((const float*)op->p[0], (const float*)op->p[1], (const float*)op->t, H, W, (int64_t*)op->p[2],
                                    (float*)op->p[3], nullptr, stream);
    case DEMFI_OP_WARP:
        if (op->bt.nb > 1)
            return demfi_warp_blend_batched(&op->a, (const float*)op->p[0], &op->b, (const float*)op->p[1], (const float*)op->p[2],
                                            (const float*)op->t, &op->o, op->nch, H, W, (float*)op->p[3], (void*)op->p[4], c->dtype, &op->bt, stream);
        if (op->p[4])
            return demfi_warp_blend_pack(&op->a, (const float*)op->p[0], &op->b, (const float*)op->p[1], (const float*)op->p[2],
                                         (const float*)op->t, &op->o, H, W, (float*)op->p[3], (void*)op->p[4], c->dtype, stream);
        return demfi_warp_blend(&op->a, (const float*)op->p[0], &op->b, (const float*)op->p[1], (const float*)op->p[2],
                                (const float*)op->t, &op->o, op->nch, H, W, (float*)op->p[3], nullptr, stream);
    }
    return demfi_set_error(DEMFI_ERR_ARG, "demfi_run_op: unknown op kind %d", op->kind);
}
