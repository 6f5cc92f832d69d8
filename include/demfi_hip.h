/*
 * demfi_hip.h -- C ABI of libdemfi_hip.so: the MI355X (gfx950) kernels of the DeMFI-Net_rb forward path.
 *
 * The upstream reference (JihyongOh/DeMFI) has NO native / FFI boundary: its hot path is the Python
 * nn.Module DeMFInet.forward (DeMFInet.py:46-179) calling stock ATen ops.  This header therefore defines
 * the boundary a host binds instead of those ATen call sites.  Each entry
 * point names the reference lines it replaces.  Rules of the ABI:
 *   - extern "C", plain pointers + sizes, no torch / C++ types;
 *   - every function returns 0 on success, a negative demfi_status otherwise, never throws;
 *     demfi_last_error() returns a thread-local message for the last failure;
 *   - the caller owns every buffer (device pointers unless stated otherwise); nothing is allocated
 *     behind the caller's back except the hipGraph objects of demfi_graph_*;
 *   - all launches go to the hipStream_t the caller passes (as void*), nothing synchronises.
 *
 * Data layouts
 *   "fat"  tensors : NHWC, element type = the path dtype (DEMFI_F16 or DEMFI_F32), described by strides;
 *   "thin" tensors : planar fp32 (the NCHW tensors the PyTorch host hands over / gets back:
 *                    flows, occlusion logits, 3-channel frames).
 * Both are described by demfi_view (element strides, so channel slices of wider buffers are views).
 */
#ifndef DEMFI_HIP_H
#define DEMFI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEMFI_ABI_VERSION 8

enum demfi_dtype { DEMFI_F16 = 0, DEMFI_F32 = 1 };

enum demfi_status {
    DEMFI_OK = 0,
    DEMFI_ERR_ARG = -1,      /* malformed descriptor / unsupported shape */
    DEMFI_ERR_HIP = -2,      /* a HIP runtime call failed (message holds hipGetErrorString) */
    DEMFI_ERR_NODEV = -3     /* no gfx950 device visible */
};

enum demfi_act { DEMFI_ACT_NONE = 0, DEMFI_ACT_RELU = 1, DEMFI_ACT_TANH = 2, DEMFI_ACT_SIGMOID = 3 };

/* Output modes of a convolution segment (v = conv + bias):
 *   STORE : dst = act(v + res)                                  (res optional)
 *   MUL   : dst = sigmoid(v) * res                              (GRU reset gate: r*h, DeMFInet.py:846-847)
 *   GRU   : dst = (1 - aux) * res + aux * tanh(v)               (GRU update, aux = z, res = h; 847-848)      */
enum demfi_mode { DEMFI_MODE_STORE = 0, DEMFI_MODE_MUL = 1, DEMFI_MODE_GRU = 2 };

/* A strided 4-D view.  Strides are in ELEMENTS of the view's own type.  ptr already points at the
 * first channel of the slice.  A NULL ptr means "absent" (or "zeros" for an input piece). */
typedef struct demfi_view {
    void*   ptr;
    int64_t sx;        /* x -> x+1        */
    int64_t sy;        /* y -> y+1        */
    int64_t sc;        /* channel -> +1   (1 for NHWC fat views) */
    int64_t sb;        /* batch image -> +1 */
    int32_t is_f32;    /* element type of ptr: 1 = fp32, 0 = fp16 */
    int32_t _pad;
} demfi_view;

/* One piece of a convolution's logical input-channel concatenation (replaces torch.cat). */
typedef struct demfi_piece {
    demfi_view v;
    int32_t nch;       /* channels taken from v (v.ptr == NULL: nch zero channels)               */
    int32_t lds_ch;    /* first channel of this piece inside its chunk                           */
    int32_t up_shift;  /* 1: the piece is read through a nearest-neighbour x2 upsample
                          (UNet decoder, DeMFInet.py:592,597,601), 0: direct                      */
    int32_t fat;       /* 1: NHWC view of the path dtype, nch*elt % 16 == 0, 16-byte vector loads;
                          0: generic element-wise loads (any strides, fp16 or fp32)               */
} demfi_piece;

#define DEMFI_MAX_PIECES 48
#define DEMFI_MAX_CHUNKS 40
#define DEMFI_MAX_SEGS    8
#define DEMFI_MAX_OCTS   32     /* cout_pad / 8, cout_pad <= 256 */

/* Input channels are consumed in chunks staged through LDS; a chunk is <= rec_bytes of channel data
 * per pixel (a multiple of one MFMA k-step = 32 bytes = 16 fp16 or 8 fp32 channels). */
typedef struct demfi_chunk {
    int32_t first_piece;
    int32_t n_pieces;
    int32_t nks;        /* k-steps in this chunk (32 bytes of channel data each)                  */
    int32_t _pad;
    int64_t w_off;      /* offset of this chunk's packed weights, in 16-byte units, cout block 0  */
} demfi_chunk;

typedef struct demfi_seg {
    demfi_view dst;
    demfi_view res;     /* optional */
    demfi_view aux;     /* optional */
    int32_t act;        /* demfi_act, STORE mode only */
    int32_t mode;       /* demfi_mode */
    int32_t scale;      /* 1, or 2 = PixelShuffle(2) store (DeMFInet.py:229): dst pixel (2y+dy, 2x+dx) */
    int32_t dy, dx;
    int32_t _pad;
} demfi_seg;

/* Implicit-GEMM convolution on the matrix cores (replaces every nn.Conv2d / nn.Conv3d(1,k,k) call site
 * of DeMFInet.py, SURVEY.md section 2.2 C1/C6-C12, together with the torch.cat / PixelShuffle /
 * UpsamplingNearest2d / activation / residual ops around them). */
typedef struct demfi_conv {
    int32_t dtype;              /* path dtype (type of fat views and packed weights)               */
    int32_t H, W;               /* OUTPUT height / width                                           */
    int32_t inH, inW;           /* input height / width as the convolution sees it (after up_shift) */
    int32_t kh, kw;
    int32_t stride;             /* 1 or 2                                                          */
    int32_t pad_y, pad_x;
    int32_t batch;              /* images sharing weights (Conv3d depth of D1 / the two FAC-FB frames) */
    int32_t cout_pad;           /* packed output channels, multiple of 32                          */
    int32_t nco;                /* 32-cout subtiles per workgroup (1..5), cout_pad % (32*nco) == 0 */
    int32_t rec_bytes;          /* LDS bytes of channel data per pixel per chunk (32/64/128)        */
    int32_t n_chunks;
    int32_t n_pieces;
    int32_t n_segs;
    int32_t cout_perm;          /* 1: the packed weights / bias are in the cout order of the persistent kernels (64-channel
                                   3x3, narrow with an NHWC destination, SepConvGRU; set by demfi_conv_build / the context for the
                                   layers those kernels own: within a
                                   32-cout subtile MFMA row r holds channel (r>>4)*16 + ((r>>2)&1)*8 + ((r>>3)&1)*4 + (r&3), so a
                                   lane's two accumulator quads are 8 consecutive channels); demfi_conv2d refuses such a
                                   descriptor when that kernel cannot take it, and an eligible one without the flag */
    int64_t w_blk_stride;       /* packed-weight stride between cout blocks, 16-byte units         */
    const void*  wpack;         /* packed weights, see demfi_pack_conv_weights                     */
    const float* bias;          /* [cout_pad] fp32 in packed cout order                            */
    const void*  zero_page;     /* >= 16 bytes of zeros in device memory (source of padding pixels for the
                                   LDS-DMA tile loads of the persistent 3x3 path); NULL disables that path */
    demfi_chunk chunks[DEMFI_MAX_CHUNKS];
    demfi_piece pieces[DEMFI_MAX_PIECES];
    demfi_seg   segs[DEMFI_MAX_SEGS];
    /* packed cout octet o = couts [8o, 8o+8): which segment, first channel inside the segment's views,
     * number of valid channels (0 = padding octet, nothing stored). */
    int32_t oct_seg[DEMFI_MAX_OCTS];
    int32_t oct_n[DEMFI_MAX_OCTS];
    int32_t oct_ch[DEMFI_MAX_OCTS];
    /* packed 32-cout subtile s: segment index when its 4 octets are one aligned run of 32 channels of an
     * NHWC view of the path dtype (dst, and res/aux when present) -> coalesced LDS-staged epilogue;
     * -1 -> per-octet direct epilogue (thin / planar / ragged outputs). */
    int32_t sub_seg[DEMFI_MAX_OCTS / 4];
    /* magic = ceil(2^32 / LW), LW = (32-1)*stride + kw: exact px / LW for px < 2^16 (filled by host) */
    uint32_t lw_magic;
    int32_t  u8_iter;           /* recursion index this descriptor belongs to (compared with demfi_u8_sink.iter) */
    /* optional uint8 sink (planar fp32 3-channel segments on the thin epilogue only, i.e. the frame-producing last layer
     * Dec_last2_2): device pointer to a demfi_u8_sink record read AT RUN TIME, so a captured hipGraph serves every
     * destination.  NULL = none. */
    const struct demfi_u8_sink* u8_sink;
    /* optional PACKED COPY of thin outputs (ABI v6; thin epilogue of the narrow persistent kernel only): besides its planar fp32
     * destination, octet o (o < 4) with pack_oct_ch[o] >= 0 also writes its channels, converted to the path dtype, as channels
     * pack_oct_ch[o] .. of the NHWC record `pack` (a lane's group of up to 4 channels is one 8-byte store; the unused channels of
     * the group are written as zeros) -- the record the next convolution stages with vector loads, i.e. what a
     * demfi_pack_planes launch over these planes would have produced (replaces the per-recursion pack of the flow / occlusion
     * deltas, DeMFInet.py:130-137 -> 800-812).  pack.ptr == NULL: none.  pack_oct_ch[o] must be a multiple of 4. */
    demfi_view pack;
    int32_t pack_oct_ch[4];
} demfi_conv;

/* uint8 egress fused into the last store: when `iter` equals the descriptor's u8_iter, segment
 * s of the convolution additionally writes crop + denorm255_np + uint8 truncation (utils.py:718-721, main.py:1165-1178:
 * float64 arithmetic) of its 3 channels to frame[s] as [h, w, 3] bytes (NULL = that frame is not wanted) and skips the
 * fp32 store of that segment. */
#define DEMFI_U8_SINK_STRIDE 256   /* bytes between the sink records of consecutive batch images (the per-t contexts' "sink" buffers) */
typedef struct demfi_u8_sink {
    uint8_t* frame[DEMFI_MAX_SEGS];
    int32_t  h, w;              /* crop (top-left h x w of the padded H x W) */
    int32_t  iter;              /* recursion index whose output is the final one (num_update - 1); -1 disables */
    int32_t  _pad;
} demfi_u8_sink;

/* ---- library / device ------------------------------------------------------------------------- */
int         demfi_abi_version(void);
const char* demfi_last_error(void);
/* Fills name[len] with the device's gcnArchName; returns DEMFI_ERR_NODEV without a GPU. */
int         demfi_device_info(char* name, int len, int* n_cu, int64_t* hbm_bytes);

/* ---- weight repack (host memory in, host memory out) -------------------------------------------
 * w_oihw : fp32 [cout, cin, kh, kw] (the state_dict tensor; Conv3d weights are passed squeezed).
 * cin_map[n_k]   : original input channel feeding packed k index (chunk-concatenated order, every
 *                  chunk padded to its k-step multiple), -1 = zero.
 * chunk_nks[n_chunks] : k-steps per chunk (sum * chans_per_kstep == n_k).
 * cout_map[cout_pad] : original output channel of packed cout, -1 = zero.
 * Packed order: [cout_blk][chunk][tap = ky*kw+kx][kstep][subtile < nco][lane < 64][16 bytes]; lane l of
 * a k-step holds, for cout = blk*32*nco + subtile*32 + (l & 31), the 8 fp16 (4 fp32) weights of packed
 * channels kstep*16 + 8*(l>>5) + j (kstep*8 + 4*(l>>5) + j) -- the A fragment of
 * v_mfma_f32_32x32x16_f16 (4 x v_mfma_f32_32x32x2_f32).
 * Returns the packed size in bytes through out_bytes (call with out == NULL to size the buffer). */
int demfi_pack_conv_weights(const float* w_oihw, int cout, int cin, int kh, int kw,
                            const int32_t* cin_map, int n_k, const int32_t* chunk_nks, int n_chunks,
                            const int32_t* cout_map, int cout_pad, int nco, int dtype,
                            void* out, int64_t* out_bytes);

/* LDS bytes one workgroup of demfi_conv2d needs for this descriptor (host-side helper). */
int64_t demfi_conv_lds_bytes(const demfi_conv* host_desc);

/* ---- kernels ----------------------------------------------------------------------------------- */
/* host_desc: descriptor in host memory (grid sizing / validation); dev_desc: the same bytes in device
 * memory (the kernel reads it through the scalar cache). */
int demfi_conv2d(const demfi_conv* host_desc, const demfi_conv* dev_desc, void* stream);

/* Fused residual block (ABI v7): y = x + conv2(relu(conv1(x))) in ONE launch -- ResBlock2D_3D / ResBlock2D (DeMFInet.py:524-563)
 * as the FAC-FB encoder (341-344), D1 (95-101: Conv3d(1,3,3) == batch 3) and D2 (158-160) use them.  h1 / h2: HOST descriptors of
 * the two 3x3 64 -> 64 fp16 convolutions exactly as demfi_conv_build makes them for demfi_conv2d (conv1: ReLU, no residual, writing
 * the intermediate; conv2: no activation, residual == conv1's input, reading the intermediate); the kernel takes the input view,
 * both packed weight sets / biases and conv2's destination from them (its arguments travel by value: no device descriptors) and
 * never touches the intermediate buffer: the intermediate lives in LDS (rounded to fp16 like the stored one, so conv1's half is
 * bit-identical to the two-launch form; conv2 accumulates onto bias + identity instead of adding the identity last: one fp32
 * rounding apart).  demfi_resblock_eligible: 1 when the pair is such a block (then demfi_resblock3x3_c64 == the two demfi_conv2d
 * launches), else 0. */
int demfi_resblock_eligible(const demfi_conv* h1, const demfi_conv* h2);
int demfi_resblock3x3_c64(const demfi_conv* h1, const demfi_conv* h2, void* stream);

/* SepConvGRU half-step with the update gate kept on chip (ABI v8; DeMFInet.py:838-857, horizontal 844-849, vertical 851-856):
 *     launch R  (demfi_gru_r)  : r * h = sigmoid(convr([h, x])) * h
 *     launch ZQ (demfi_gru_zq) : h' = (1 - z) h + z tanh(convq([r * h, x])),  z = sigmoid(convz([h, x]))
 * instead of the z | r and q launches of demfi_conv2d: z never goes to memory (handed from the z waves to the q waves through LDS,
 * rounded to fp16 as the stored z was), 896 instead of 1 152 bytes per pixel and half-step.  The arguments are HOST descriptors of
 * the plain 64-cout layers exactly as demfi_conv_build makes them for demfi_conv2d (1x5 or 5x1, fp16, two 64-channel NHWC pieces):
 *   hr: [h, x] -> MUL epilogue with res == h;   hz: [h, x] -> sigmoid, STORE into the z buffer (never touched by the fused launch);
 *   hq: [r*h, x] -> GRU epilogue with res == h, aux == hz's destination.
 * The kernel arguments travel by value (no device descriptors).  *_eligible: 1 when the descriptors are such layers (then the fused
 * launch equals the demfi_conv2d launches up to the summation order of the fp32 accumulators), else 0. */
int demfi_gru_r_eligible(const demfi_conv* hr);
int demfi_gru_r(const demfi_conv* hr, void* stream);
int demfi_gru_zq_eligible(const demfi_conv* hz, const demfi_conv* hq);
int demfi_gru_zq(const demfi_conv* hz, const demfi_conv* hq, void* stream);

/* pixel_reshuffle(cat(B0,B1,B-1,B2), 2) (DeMFInet.py:234-235, 290-316): x fp32 [3,4,H,W] (C,T order of the
 * module input, batch 1) -> fat NHWC [H/2, W/2, 48], channel = (frame*3 + c)*4 + ry*2 + rx. */
int demfi_space_to_depth(const float* x, void* out, int dtype, int H, int W, void* stream);

/* Reflect padding of the harness (utils.py:1360-1365): x [planes,h,w] -> out [planes,H,W], bottom/right. */
int demfi_reflect_pad(const float* x, float* out, int planes, int h, int w, int H, int W, void* stream);

/* torch.mean(x[:, :, 0:2], dim=2) (DeMFInet.py:178): x [3,4,H,W] -> out [3,H,W]. */
int demfi_overlay_mean(const float* x, float* out, int H, int W, void* stream);

/* CFR_flow_t_align (DeMFInet.py:606-622) = two forward splats (fwarp 625-671, sample_one 683-729) +
 * the linear combination / normalisation.  flow01, flow10: planar fp32 [2,H,W]; t: device pointer to
 * one fp32; acc: caller workspace of demfi_cfr_workspace_bytes(H, W) bytes (six int64 [H,W] planes + per-tile flags)
 * that MUST be all-zero on entry (allocate it zeroed once) and is left all-zero on exit; out: planar fp32 [4,H,W] =
 * (flow_t0, flow_t1).  The splat accumulates in 64-bit fixed point (2^-32 resolution) so the result
 * does not depend on the order of the adds (LDS atomics per target tile; global atomics only for sources displaced
 * by more than 32 px).  dbg_idx (optional, may be NULL): int32
 * [2 flows][4 corners][H*W] flat target index of every source pixel, -1 where masked off
 * (sample_one's ids / mask, DeMFInet.py:712-719) for the index-parity tests. */
int64_t demfi_cfr_workspace_bytes(int H, int W);
int demfi_cfr_reset(int64_t* acc, int H, int W, void* stream);      /* re-zero the workspace after an aborted launch */
int demfi_cfr_flow_align(const float* flow01, const float* flow10, const float* t, int H, int W,
                         int64_t* acc, float* out, int32_t* dbg_idx, void* stream);

/* Eq.(2): two backward warps (bwarp, DeMFInet.py:732-766) blended with the occlusion map
 * (DeMFInet.py:66-71, 90-93, 146-149):
 *   o0 = sigmoid(logit); out = ((1-t) o0 bwarp(A,fa) + t (1-o0) bwarp(B,fb)) / ((1-t) o0 + t (1-o0)).
 * A, B, out: views with C channels (fat NHWC of the path dtype, or thin planar fp32); fa, fb: planar
 * fp32 [2,H,W]; logit: planar fp32 [H,W]; t: device fp32.  occ_out (optional): sigmoid(logit) [H,W].
 * dbg_maps (optional): int32 [2 warps][3][H*W] = floor x index, floor y index, bit0-3 in-bounds of
 * (nw,ne,sw,se) | bit4 validity mask -- the integer maps of grid_sample for the index-parity tests. */
int demfi_warp_blend(const demfi_view* A, const float* fa, const demfi_view* B, const float* fb,
                     const float* logit, const float* t, const demfi_view* out, int C, int H, int W,
                     float* occ_out, int32_t* dbg_maps, void* stream);

/* The same for 3-channel planar frames (the PWB of the recursion, DeMFInet.py:140-149) with the packed copy the next layer
 * reads: pack8 = NHWC [H,W,8] of pack_dtype holding [out 0..2 | fa | fb | sigmoid(logit)] (Agg3's per-recursion planes,
 * DeMFInet.py:151-155) -- replaces a demfi_pack_planes launch. */
int demfi_warp_blend_pack(const demfi_view* A, const float* fa, const demfi_view* B, const float* fb,
                          const float* logit, const float* t, const demfi_view* out, int H, int W, float* occ_out,
                          void* pack8, int pack_dtype, void* stream);

/* ---- batched point-wise launches (ABI v5): the same op for the nb per-t contexts of one trunk set in ONE launch ----------------
 * The per-t buffers of a context set are laid out tensor-major (copy c of a buffer sits c * stride bytes behind copy 0), so a
 * batched launch takes the pointers of context 0 plus one BYTE stride per pointer (0 = a window-level buffer every context shares:
 * trunk features, flow_01 / flow_10).  Counterpart of the reference running its whole forward once per t (main.py:1121-1178):
 * the t loop moves inside the kernel.  For the fat warp (Ft = blend of the two warped trunk feature maps, DeMFInet.py:66-71) the
 * contexts are the INNERMOST loop of a tile, so F0 / F1 are fetched from HBM once per window instead of once per t. */
typedef struct demfi_batch {
    int32_t nb;            /* contexts in the launch; 0 or 1 = a plain launch                                              */
    int32_t _pad;          /* fat warp only: 0 = the contexts are the innermost loop of a tile, 1 = one grid slice per context (the
                              same work as nb launches without their gaps and tails; round 5)                                  */
    int64_t a, b, o;       /* byte strides of the op's A / B / out views                                                   */
    int64_t t;             /* ... of the device time instant                                                               */
    int64_t p[32];         /* ... of the op's pointer arguments, in demfi_op.p order                                       */
} demfi_batch;
/* p order: fa, fb, logit, occ_out, pack8 (demfi_op of kind WARP) */
int demfi_warp_blend_batched(const demfi_view* A, const float* fa, const demfi_view* B, const float* fb, const float* logit,
                             const float* t, const demfi_view* out, int C, int H, int W, float* occ_out, void* pack8,
                             int pack_dtype, const demfi_batch* bt, void* stream);
/* p order: flow01, flow10, acc, out (kind CFR) */
int demfi_cfr_flow_align_batched(const float* flow01, const float* flow10, const float* t, int H, int W, int64_t* acc, float* out,
                                 const demfi_batch* bt, void* stream);
/* The same with the packed copy the next layer reads (ABI v8; p order of a batched launch: flow01, flow10, acc, out, logit, pack16): the finish
 * also writes pack16 = NHWC [H,W,16] of pack_dtype holding [flow_t0, flow_t1 | flow_01, flow_10, logit | 7 zeros] -- the thin members of Agg1
 * (DeMFInet.py:77) as Refine_Module.enc1 stages them -- replacing a demfi_pack_planes launch per window.  logit: planar fp32 [H,W]; bt may be NULL. */
int demfi_cfr_flow_align_pack(const float* flow01, const float* flow10, const float* logit, const float* t, int H, int W, int64_t* acc,
                              float* out, void* pack16, int pack_dtype, const demfi_batch* bt, void* stream);

/* bilinear_sampler at ABSOLUTE flow coordinates (FGAC, DeMFInet.py:413-419, 499-514; rr = sr = 0):
 * src, out fat views with C channels; flow planar fp32 [2,H,W]. */
int demfi_fgac_gather(const demfi_view* src, const float* flow, const demfi_view* out, int C, int H,
                      int W, int32_t* dbg_maps, void* stream);

/* Generalised FGAC (radii rr > 0, DeMFInet.py:401-445): correlation of the (2rr+1)^2 bilinear samples of ref_k with
 * source_k over the C = 64 channels, softmax over the window, attention-weighted sum (Eq. 3).  fp16 NHWC views.
 * mode 0: the index map the reference code computes when its hard-coded radii are overridden (pinned by fixtures from a
 * patched in-memory copy of the reference); mode 1: window centred on flow[y,x] (the paper's description), (2rr+2)^2
 * texel window staged per pixel in LDS; both: wavefront-shuffle channel reduction + softmax.  attn_out (optional):
 * fp32 [(2rr+1)^2, H, W] softmax weights.  sr > 0 (DeMFInet.py:417, 434): apply demfi_avg_pool_fat to ref_k / source_k
 * first.  rr in {1, 2}. */
int demfi_fgac_window(const demfi_view* ref_k, const demfi_view* source_k, const float* flow, const demfi_view* out,
                      int C, int H, int W, int rr, int mode, float* attn_out, void* stream);
/* F.avg_pool2d(x, 2sr+1, stride 1, padding sr) (count_include_pad), fp16 NHWC. */
int demfi_avg_pool_fat(const demfi_view* src, const demfi_view* out, int C, int H, int W, int sr, void* stream);

/* Visualisation extras of FGAC.forward (DeMFInet.py:454-496; returned by DeMFInet.forward when args.visualization_flag or is_training,
 * 167-176).  demfi_absmean_map: out[H,W] = mean over the C channels of |a| (b == NULL) or |a - b| (torch.mean(torch.abs(.), 1));
 * demfi_minmax_normalize: plane = (plane - min(plane)) / max(plane - min(plane)) in place, the reference's two in-place steps
 * (459-461), deterministic; scratch: demfi_minmax_scratch_floats() floats; demfi_one_minus: out = 1 - in (the (1 - w_sr) map, 495). */
int     demfi_absmean_map(const demfi_view* a, const demfi_view* b, float* out, int C, int H, int W, void* stream);
int64_t demfi_minmax_scratch_floats(void);
int     demfi_minmax_normalize(float* plane, int64_t n, float* scratch, void* stream);
int     demfi_one_minus(const float* in, float* out, int64_t n, void* stream);

/* Eq.(4) gate blend (DeMFInet.py:452): out = w*source + (1-w)*e; w planar fp32 [H,W]. */
int demfi_gate_blend(const float* w, const demfi_view* source, const demfi_view* e, const demfi_view* out,
                     int C, int H, int W, void* stream);

/* Pack planar fp32 channels into an NHWC slice of the path dtype (replaces the thin members of torch.cat at
 * DeMFInet.py:77, 117-120, 123, 151-155 for the consuming convolutions).  planes: HOST array of nch device
 * pointers to [H,W] fp32 planes (NULL = zero channel); nch multiple of 8, <= 32; dst: first channel of the slice,
 * dst_pix_stride in elements. */
int demfi_pack_planes(const float* const* planes, int nch, void* dst, int dtype, int64_t dst_pix_stride,
                      int H, int W, void* stream);
/* batched over contexts: bt->p[i] = byte stride of plane i, bt->o = byte stride of dst */
int demfi_pack_planes_batched(const float* const* planes, int nch, void* dst, int dtype, int64_t dst_pix_stride, int H, int W,
                              const demfi_batch* bt, void* stream);

/* uint8 frame I/O of the boundary caller.
 * demfi_u8_to_window: frames = HOST array of 4 device pointers to BGR uint8 [h,w,3] images in the module's frame order
 * (B0,B1,B-1,B2); writes x [3,4,H,W] fp32 = RGBframes_np2Tensor normalisation (utils.py:224-238) + reflect padding to
 * H x W (utils.py:1363).
 * demfi_frame_to_u8: frame planar fp32 [3,H,W] -> out uint8 [h,w,3]: crop + denorm255_np (utils.py:718-721) + uint8
 * truncation (main.py:1165-1178), bit-identical to the reference's float64 arithmetic. */
int demfi_u8_to_window(const uint8_t* const* frames, int h, int w, float* x, int H, int W, void* stream);
/* The same fused with the first layers' loads: one pass over the 4 uint8 frames writes x (fp32 [3,4,H,W], still needed by
 * the Mixer / D2 inputs), the space-to-depth tensor of FF_RDB (NHWC [H/2,W/2,48], path dtype; pixel_reshuffle,
 * DeMFInet.py:234-235) and the overlay (mean of B0, B1; DeMFInet.py:178). */
int demfi_u8_ingest(const uint8_t* const* frames, int h, int w, float* x, void* s2d, float* overlay, int dtype, int H, int W,
                    void* stream);
/* one BGR uint8 [h,w,3] frame -> planar fp32 [3,h,w] with the same arithmetic (ground-truth frames of the evaluation) */
int demfi_u8_to_planar(const uint8_t* frame, int h, int w, float* out, void* stream);
int demfi_frame_to_u8(const float* frame, uint8_t* out, int h, int w, int H, int W, void* stream);

/* ---- on-GPU evaluation ------------------------------------------------------------
 * psnr (utils.py:652-660) and MATLAB-style 11x11 Gaussian ssim (utils.py:663-705) of one predicted frame against its
 * target as test() computes them (main.py:762-770): pred is rounded after denorm255_np, the target is not (round_gt = 0)
 * or is (round_gt = 1); fp64 arithmetic.  pred / gt: planar fp32 [3, ., .] in [-1,1] with explicit row / channel strides
 * (elements), so a crop of a padded buffer is a view.  workspace: demfi_eval_workspace_bytes(h, w) bytes;
 * out3 (device): {psnr dB (inf when identical), ssim, mse}.  Deterministic (fixed-order reductions). */
int64_t demfi_eval_workspace_bytes(int h, int w);
int demfi_eval_frame(const float* pred, int64_t pred_row_stride, int64_t pred_ch_stride, const float* gt,
                     int64_t gt_row_stride, int64_t gt_ch_stride, int h, int w, int round_gt, double* workspace,
                     double* out3, void* stream);

/* ---- PNG codec of the clip I/O edge (host only, thread-safe) -------------------------
 * Counterpart of cv2.imread (utils.py:583-593) / cv2.imwrite (main.py:1165-1178) on zlib: uint8 [h,w,3] images in cv2's
 * B,G,R order, `stride` bytes per row.  decode: 8/16-bit gray / RGB / palette / +alpha, non-interlaced -> BGR8 (what
 * cv2.imread(path) returns); encode: 8-bit RGB, zlib `level` 0..9, filter -1 = adaptive (libpng's heuristic) or 0..4 (1 = Sub, OpenCV's default),
 * strategy -1 = Z_RLE at level <= 3 (OpenCV's default) or a zlib strategy constant. */
int64_t demfi_png_encode_bound(int h, int w);
int demfi_png_encode(const uint8_t* bgr, int h, int w, int64_t stride, int level, int filter, int strategy, uint8_t* out,
                     int64_t out_cap, int64_t* out_bytes);
int demfi_png_info(const uint8_t* data, int64_t n, int* h, int* w);
int demfi_png_decode(const uint8_t* data, int64_t n, uint8_t* bgr, int64_t stride, int h_expect, int w_expect);

/* ---- convolution descriptor builder (host only) ---------------------------------------------------
 * The ONE implementation of the layout logic behind demfi_conv: packs the logical input-channel concatenation
 * (replaces torch.cat) into LDS chunks, routes output channels to destination slices, repacks the weights.
 * Two-call pattern: with wpack == NULL only *wpack_bytes and *cout_pad are written. */
typedef struct demfi_conv_src {
    demfi_view v;           /* first channel of the piece                                               */
    int32_t fat;            /* 1: NHWC view of the path dtype (vector loads), 0: generic (planar fp32 ...) */
    int32_t up_shift;       /* 1: read through a nearest-neighbour x2 upsample                          */
    int32_t nch;            /* channels of the piece                                                    */
    int32_t _pad;
    const int32_t* cin;     /* [nch] original input channel of each channel, -1 = unused (zero weights)  */
} demfi_conv_src;

typedef struct demfi_conv_dst {
    demfi_view dst, res, aux;   /* res / aux: ptr == NULL when absent                                   */
    int32_t act, mode, scale, dy, dx;
    int32_t n;              /* output channels routed to this destination                               */
    const int32_t* couts;   /* [n] original output channel of channel j of the destination slice        */
} demfi_conv_dst;

/* desc: filled except wpack / bias / zero_page (the caller places the blobs and sets the pointers).
 * wpack: *wpack_bytes bytes; bias_packed: *cout_pad floats. H, W: OUTPUT size. */
int demfi_conv_build(int dtype, int H, int W, int stride, int batch, const float* w_oihw, const float* bias,
                     int cout, int cin, int kh, int kw, const demfi_conv_src* srcs, int n_srcs,
                     const demfi_conv_dst* dsts, int n_dsts, demfi_conv* desc, void* wpack, int64_t* wpack_bytes,
                     float* bias_packed, int32_t* cout_pad);

/* ---- forward context: the launch plan of DeMFInet.forward (DeMFInet.py:46-179) behind the C ABI -------
 * A context owns NO device memory: the caller allocates demfi_workspace_bytes() bytes (zero-filled) and binds them.
 * Inside the workspace: [packed weights + biases (one flat blob: the buffer a multi-GPU launch may broadcast) |
 * descriptors | n_trunk trunk buffer sets | n_trunk * n_ctx per-t buffer sets].  Since ABI v7 the big activation buffers of a set
 * are planned by liveness and share an arena (buffers that are never alive together occupy the same bytes; the workspace of the
 * 720p x8 configuration with 3 x 7 sets: 87.5 -> 36.1 GB): the named inputs / outputs of demfi_ctx_buffer ("x", "t", "sink",
 * "finals", "delta", "occ", "sharp1", "overlay", "ffo", "aF", "F01", "ft", "gate", "viz") own their memory, other buffers hold their value only
 * while the plan needs it.  Everything a per-t context touches lies in memory no other context touches (slots with a common context
 * stride), so the independence promised below is kept.  The liveness plan holds for executions in PLAN ORDER (trunk, head, recursions
 * 0..N-1, or prefixes of it): demfi_run_op on a single op out of order reads memory later tenants have recycled.  The scratch of a fused
 * launch (DEMFI_OP_RESBLOCK's intermediate, DEMFI_OP_GRU_ZQ's z buffer) has NO memory under the arena: such an op must not be run as its
 * constituent convolutions on the bound workspace, and demfi_ctx_bind fails if it would fuse other launches than the sizing pass did.
 * Environment DEMFI_ARENA=0: one region per buffer (debugging, isolated per-op profiling).  n_trunk / n_ctx > 1 build
 * independent buffer sets so that a scheduler can overlap the trunk of window w+1 with the time instants of window w,
 * and several time instants of one window on different streams (results do not depend on it).
 * Re-entrant per context; one context per (GPU, frame size, dtype). */
typedef struct demfi_ctx demfi_ctx;

typedef struct demfi_hparams {           /* DeMFInet.py:17-21, 32, 42, 326, 328; main.py:88-101 defaults */
    int32_t nf;                          /* 64 (only value the kernels are built for)                    */
    int32_t scale_factor;                /* 2                                                            */
    int32_t num_resb_facfb;              /* 5                                                            */
    int32_t num_resb_dec;                /* 5                                                            */
    int32_t shared_fgac;                 /* 1                                                            */
    int32_t fgac_rr, fgac_sr;            /* 0, 0: the radii hard-coded at DeMFInet.py:401-402 (generalised FGAC when > 0) */
    int32_t flags;                       /* bit 0: generalised FGAC index map (0 = as the reference code computes it, 1 = pixel-centred);
                                            bit 1 (DEMFI_HP_EXTRAS, ABI v8): also compute the maps of the reference's visualisation /
                                            training return tuples (DeMFInet.py:167-176, 454-496) into the trunk buffer "viz"        */
} demfi_hparams;
#define DEMFI_HP_FGAC_CENTRED 1
#define DEMFI_HP_EXTRAS       2

enum demfi_op_kind {
    DEMFI_OP_CONV = 0, DEMFI_OP_PACK = 1, DEMFI_OP_S2D = 2, DEMFI_OP_OVERLAY = 3, DEMFI_OP_FGAC = 4, DEMFI_OP_GATE = 5,
    DEMFI_OP_CFR = 6, DEMFI_OP_WARP = 7, DEMFI_OP_FGAC_WINDOW = 8, DEMFI_OP_AVG_POOL = 9,
    DEMFI_OP_RESBLOCK = 10,     /* fused residual block: conv = descriptor of conv1, nch = descriptor of conv2 (demfi_resblock3x3_c64) */
    DEMFI_OP_GRU_R = 11,        /* reset gate of a SepConvGRU half-step: conv = descriptor of convr (demfi_gru_r)                        */
    DEMFI_OP_GRU_ZQ = 12,       /* update gate + candidate + blend: conv = descriptor of convz, nch = descriptor of convq (demfi_gru_zq) */
    DEMFI_OP_VIZ = 13           /* visualisation extras (DEMFI_HP_EXTRAS): conv = 0 channel mean of |a - b| (b optional) -> plane p[0];
                                   1 min-max normalisation of plane p[0] in place (scratch p[1]); 2 plane p[0] = 1 - plane p[1]           */
};
enum demfi_segment { DEMFI_SEG_TRUNK = 0, DEMFI_SEG_T_HEAD = 1, DEMFI_SEG_ITER = 2,     /* TRUNK: ops 0, 1 = s2d, overlay (the prologue demfi_ingest_u8 replaces) */
                     DEMFI_SEG_TB_HEAD = 3, DEMFI_SEG_TB_ITER = 4 };                     /* the batched per-t plan of demfi_forward_tb (context index ignored) */

/* One launch of the plan (introspection for tests / per-launch profiling; pointers are already bound). */
typedef struct demfi_op {
    int32_t kind;           /* demfi_op_kind                                                             */
    int32_t conv;           /* CONV: descriptor index (demfi_ctx_conv_desc); FGAC_WINDOW: rr; AVG_POOL: sr; RESBLOCK: conv1's descriptor */
    int32_t nch;            /* PACK: channels (multiple of 8); WARP / FGAC / GATE: C; RESBLOCK: conv2's descriptor */
    int32_t _pad;           /* FGAC_WINDOW: index map (0 reference code, 1 pixel-centred)                */
    int64_t macs;           /* CONV: algorithmic multiply-accumulates of the launch                      */
    demfi_view a, b, o;     /* WARP: A, B, out; FGAC: src, -, out; GATE: source, e, out; PACK: o = dst     */
    const void* p[32];      /* PACK: plane pointers; WARP: fa, fb, logit, occ_out, pack8 (or NULL); FGAC: flow; GATE: w;
                               CFR: flow01, flow10, acc, out; S2D / OVERLAY: x, out; all: t where needed  */
    const void* t;          /* device fp32 time instant (CFR, WARP)                                      */
    char name[64];
    demfi_batch bt;         /* batched plan: nb > 1 = ONE launch for the nb per-t contexts (CFR, WARP, PACK)   */
} demfi_op;

int     demfi_ctx_create(int H, int W, int max_updates, int dtype, const demfi_hparams* hp /* NULL = defaults */,
                         int n_trunk, int n_ctx, demfi_ctx** out);
int     demfi_ctx_destroy(demfi_ctx* ctx);
/* name: state_dict key ("FF_RDB_Module.SFENet1.weight" ...), host fp32, shape as in the state_dict (Conv3d weights
 * [cout,cin,1,3,3] accepted).  All 260 (270 non-shared) tensors must be loaded before demfi_ctx_bind. */
int     demfi_load_weight(demfi_ctx* ctx, const char* name, const float* host, const int64_t* shape, int ndim);
int64_t demfi_ctx_workspace_bytes(const demfi_ctx* ctx);
/* Size without a context: same number as a context created with these arguments reports. */
int64_t demfi_workspace_bytes(int H, int W, int max_updates, int dtype, int n_trunk, int n_ctx);
/* Builds the plan into `workspace` (device memory, zero-filled by the caller; on_host != 0: host memory, nothing is
 * launched -- the CPU plan tests) and uploads weights + descriptors on `stream` (synchronises it before returning). */
int     demfi_ctx_bind(demfi_ctx* ctx, void* workspace, int64_t bytes, int on_host, void* stream);
/* Region of the workspace holding the packed weights (offset, bytes): identical on every rank after a broadcast. */
int     demfi_ctx_weight_region(const demfi_ctx* ctx, int64_t* offset, int64_t* bytes);
/* Named buffer of trunk context `trunk` / per-t context `c` (c = -1: trunk buffers): byte offset inside the workspace,
 * element kind (0 path dtype NHWC [B,h,w,C], 1 fp32 planar [C,h,w], 2 int64 raw) and dims[4].  Names: "x" (input
 * [3,4,H,W] fp32), "overlay", "ffo"; per-t: "t", "sharp1" (S0',S1',St' = D1 frames, 9 planes), "finals" ([N][3 frames][3]),
 * "delta" ([N+1][5]: flow_t0, flow_t1, occ logit), "occ" ([N+1]) ... every buffer of the plan is addressable. */
int     demfi_ctx_buffer(const demfi_ctx* ctx, int trunk, int c, const char* name, int64_t* offset, int32_t* kind,
                         int32_t dims[4]);
/* uint8 boundary of a context.  demfi_ingest_u8: 4 BGR uint8 [h,w,3] device frames (B0,B1,
 * B-1,B2; HOST array of 4 device pointers) -> the context's x / s2d / overlay buffers (normalise + reflect pad + pixel
 * reshuffle in one kernel); follow it with demfi_forward_trunk_body.  The per-t context's buffer "sink" is a
 * demfi_u8_sink record: fill it (device memory) before demfi_forward_t and the last layer writes uint8 frames directly. */
int     demfi_ingest_u8(demfi_ctx* ctx, int trunk, const uint8_t* const* frames, int h, int w, void* stream);
int     demfi_forward_trunk_body(demfi_ctx* ctx, int trunk, void* stream);
/* t-independent segment (FF_RDB + FAC-FB, DeMFInet.py:59, 74) of trunk context `trunk`; x: device fp32 [3,4,H,W]
 * copied into the context's input buffer first, or NULL when the caller already filled buffer "x". */
int     demfi_forward_trunk(demfi_ctx* ctx, int trunk, const float* x, void* stream);
/* per-t segment (DeMFInet.py:63-165) of per-t context c reading trunk context `trunk`; t is read from buffer "t". */
int     demfi_forward_t(demfi_ctx* ctx, int trunk, int c, int n_updates, void* stream);
/* The same per-t segment for ALL n_ctx per-t contexts of trunk set `trunk` as one launch sequence: every convolution runs
 * once over batch x n_ctx images (the copies of a per-t buffer are contiguous in the workspace: copy c of buffer b sits
 * c * stride(b) bytes behind copy 0, see demfi_ctx_buffer), the point-wise kernels once per context.  Context c reads its
 * own "t" and "sink" buffers; results are bit-identical to n_ctx calls of demfi_forward_t.  The time instants of a window
 * are independent given the trunk (DeMFInet.py:63-165 runs once per t in the reference's loop, main.py:1121-1178): at
 * x8 the 7 instants of a window are one batch, which takes the launch tails / pipeline fill of the small per-t grids
 * out of the picture.  n_ctx >= 2. */
int     demfi_forward_tb(demfi_ctx* ctx, int trunk, int n_updates, void* stream);
/* demfi_forward_tb for a consumer of the LAST recursion's frames only -- which is what the reference's inference pipelines
 * are (utils.py:1430-1434 "considering list of final", main.py:798, 1153: Sharps_final[-1]).  The pixel-flow warp + D2 tail of
 * a recursion (DeMFInet.py:146-165) feeds nothing but that recursion's Sharps_final entry (the recursion state is F_rec and the
 * flow / occlusion logits, 130-137), so it is run for recursion n_updates - 1 only: "finals" / "occ" of the earlier recursions
 * are NOT written, everything else (flows, logits, the final frames, the uint8 sink) is bit-identical to demfi_forward_tb. */
int     demfi_forward_tb_final(demfi_ctx* ctx, int trunk, int n_updates, void* stream);
/* introspection / per-launch execution */
int     demfi_ctx_num_ops(const demfi_ctx* ctx, int segment, int trunk, int c, int iter);
int     demfi_ctx_get_op(const demfi_ctx* ctx, int segment, int trunk, int c, int iter, int index, demfi_op* out);
int     demfi_ctx_num_convs(const demfi_ctx* ctx);
const demfi_conv* demfi_ctx_conv_desc(const demfi_ctx* ctx, int index);          /* host copy */
int     demfi_run_op(demfi_ctx* ctx, const demfi_op* op, void* stream);

/* ---- single-call operators behind the C ABI (ABI v7; SURVEY.md section 8b: demfi_gru_sep / demfi_fgac) -----------------------------
 * SepConvGRU (DeMFInet.py:827-857: h' = GRU_5x1(GRU_1x5(h, x), x)) and FGAC at its hard-coded radii rr = sr = 0 (DeMFInet.py:361-452:
 * conv_ref_k -> bilinear sample at the absolute flow coordinates -> fusion -> gate -> w source + (1 - w) E_s) as contexts of their own,
 * for a host that swaps ONE module of the reference network.  An operator context is a demfi_ctx whose only segment is the operator:
 * the whole context API applies -- demfi_load_weight with the reference module's own keys ("convz1.weight" ... "convq2.bias";
 * "conv_ref_k.*", "conv_source_k.*" (accepted, dead at rr = 0), "fusion.*", "w_gen.*", "w_gen_2.*"), demfi_ctx_workspace_bytes,
 * demfi_ctx_bind (caller-owned zero-filled workspace), demfi_ctx_buffer(ctx, 0, -1, name, ...) for the NHWC [batch,H,W,64] buffers of the
 * path dtype ("h", "x", "out" / "ref", "source", "out") and the planar fp32 ones ("flow" [batch,2,H,W], "w" [batch,1,H,W] = the gate),
 * demfi_ctx_num_ops / demfi_ctx_get_op on DEMFI_SEG_TRUNK, demfi_ctx_destroy -- and demfi_operator_run launches it (4 convolutions
 * for the GRU: z | r fused and q per direction; 4 convolutions + one gather and one blend per image for FGAC). */
int demfi_gru_sep_create(int batch, int H, int W, int dtype, demfi_ctx** out);
int demfi_fgac_create(int batch, int H, int W, int dtype, demfi_ctx** out);
int demfi_operator_run(demfi_ctx* ctx, void* stream);

/* ---- hipGraph capture of a launch sequence ------------------------------------------------------ */
int demfi_graph_begin(void* stream);
int demfi_graph_end(void* stream, void** graph_exec_out);
int demfi_graph_launch(void* graph_exec, void* stream);
int demfi_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* DEMFI_HIP_H */
