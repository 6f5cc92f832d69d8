"""ctypes binding of libdemfi_hip.so (include/demfi_hip.h).

The HIP library is the product; there is no CPU fallback.  ``load()`` raises ``RuntimeError`` when the
shared object is missing, and every call that returns a negative status raises with the library's message.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DEMFI_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libdemfi_hip.so')   # override: ablation builds

F16, F32 = 0, 1
ABI_VERSION = 8
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
MODE_STORE, MODE_MUL, MODE_GRU = 0, 1, 2
MAX_PIECES, MAX_CHUNKS, MAX_SEGS, MAX_OCTS = 48, 40, 8, 32


class View(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('sx', C.c_int64), ('sy', C.c_int64), ('sc', C.c_int64), ('sb', C.c_int64),
                ('is_f32', C.c_int32), ('_pad', C.c_int32)]


class Piece(C.Structure):
    _fields_ = [('v', View), ('nch', C.c_int32), ('lds_ch', C.c_int32), ('up_shift', C.c_int32), ('fat', C.c_int32)]


class Chunk(C.Structure):
    _fields_ = [('first_piece', C.c_int32), ('n_pieces', C.c_int32), ('nks', C.c_int32), ('_pad', C.c_int32),
                ('w_off', C.c_int64)]


class Seg(C.Structure):
    _fields_ = [('dst', View), ('res', View), ('aux', View), ('act', C.c_int32), ('mode', C.c_int32),
                ('scale', C.c_int32), ('dy', C.c_int32), ('dx', C.c_int32), ('_pad', C.c_int32)]


class Conv(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('inH', C.c_int32), ('inW', C.c_int32),
                ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32), ('pad_y', C.c_int32),
                ('pad_x', C.c_int32), ('batch', C.c_int32), ('cout_pad', C.c_int32), ('nco', C.c_int32),
                ('rec_bytes', C.c_int32), ('n_chunks', C.c_int32), ('n_pieces', C.c_int32), ('n_segs', C.c_int32),
                ('cout_perm', C.c_int32), ('w_blk_stride', C.c_int64), ('wpack', C.c_void_p), ('bias', C.c_void_p), ('zero_page', C.c_void_p),
                ('chunks', Chunk * MAX_CHUNKS), ('pieces', Piece * MAX_PIECES), ('segs', Seg * MAX_SEGS),
                ('oct_seg', C.c_int32 * MAX_OCTS), ('oct_n', C.c_int32 * MAX_OCTS), ('oct_ch', C.c_int32 * MAX_OCTS),
                ('sub_seg', C.c_int32 * (MAX_OCTS // 4)),
                ('lw_magic', C.c_uint32), ('u8_iter', C.c_int32), ('u8_sink', C.c_void_p),
                ('pack', View), ('pack_oct_ch', C.c_int32 * 4)]


class U8Sink(C.Structure):
    _fields_ = [('frame', C.c_void_p * MAX_SEGS), ('h', C.c_int32), ('w', C.c_int32), ('iter', C.c_int32), ('_pad', C.c_int32)]


class ConvSrc(C.Structure):
    _fields_ = [('v', View), ('fat', C.c_int32), ('up_shift', C.c_int32), ('nch', C.c_int32), ('_pad', C.c_int32),
                ('cin', C.POINTER(C.c_int32))]


class ConvDst(C.Structure):
    _fields_ = [('dst', View), ('res', View), ('aux', View), ('act', C.c_int32), ('mode', C.c_int32), ('scale', C.c_int32),
                ('dy', C.c_int32), ('dx', C.c_int32), ('n', C.c_int32), ('couts', C.POINTER(C.c_int32))]


class HParams(C.Structure):
    _fields_ = [('nf', C.c_int32), ('scale_factor', C.c_int32), ('num_resb_facfb', C.c_int32), ('num_resb_dec', C.c_int32),
                ('shared_fgac', C.c_int32), ('fgac_rr', C.c_int32), ('fgac_sr', C.c_int32), ('flags', C.c_int32)]


class Batch(C.Structure):
    """demfi_batch: one launch for nb per-t contexts; byte strides between the contexts' copies of every pointer."""
    _fields_ = [('nb', C.c_int32), ('_pad', C.c_int32), ('a', C.c_int64), ('b', C.c_int64), ('o', C.c_int64), ('t', C.c_int64),
                ('p', C.c_int64 * 32)]


class Op(C.Structure):
    _fields_ = [('kind', C.c_int32), ('conv', C.c_int32), ('nch', C.c_int32), ('_pad', C.c_int32), ('macs', C.c_int64),
                ('a', View), ('b', View), ('o', View), ('p', C.c_void_p * 32), ('t', C.c_void_p), ('name', C.c_char * 64),
                ('bt', Batch)]


_SIGS = {
    'demfi_abi_version': (C.c_int, []),
    'demfi_last_error': (C.c_char_p, []),
    'demfi_device_info': (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    'demfi_pack_conv_weights': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.POINTER(C.c_int64)]),
    'demfi_conv_lds_bytes': (C.c_int64, [C.POINTER(Conv)]),
    'demfi_conv2d': (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_void_p]),
    'demfi_resblock_eligible': (C.c_int, [C.POINTER(Conv), C.POINTER(Conv)]),
    'demfi_resblock3x3_c64': (C.c_int, [C.POINTER(Conv), C.POINTER(Conv), C.c_void_p]),
    'demfi_gru_r_eligible': (C.c_int, [C.POINTER(Conv)]),
    'demfi_gru_r': (C.c_int, [C.POINTER(Conv), C.c_void_p]),
    'demfi_gru_zq_eligible': (C.c_int, [C.POINTER(Conv), C.POINTER(Conv)]),
    'demfi_gru_zq': (C.c_int, [C.POINTER(Conv), C.POINTER(Conv), C.c_void_p]),
    'demfi_space_to_depth': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'demfi_reflect_pad': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'demfi_overlay_mean': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'demfi_cfr_workspace_bytes': (C.c_int64, [C.c_int, C.c_int]),
    'demfi_cfr_reset': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'demfi_cfr_flow_align': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    'demfi_warp_blend': (C.c_int, [C.POINTER(View), C.c_void_p, C.POINTER(View), C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(View), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'demfi_warp_blend_pack': (C.c_int, [C.POINTER(View), C.c_void_p, C.POINTER(View), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(View), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'demfi_warp_blend_batched': (C.c_int, [C.POINTER(View), C.c_void_p, C.POINTER(View), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(View), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Batch),
                                           C.c_void_p]),
    'demfi_cfr_flow_align_batched': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                               C.POINTER(Batch), C.c_void_p]),
    'demfi_cfr_flow_align_pack': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.POINTER(Batch), C.c_void_p]),
    'demfi_pack_planes_batched': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(Batch),
                                            C.c_void_p]),
    'demfi_fgac_gather': (C.c_int, [C.POINTER(View), C.c_void_p, C.POINTER(View), C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p]),
    'demfi_fgac_window': (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_void_p, C.POINTER(View), C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'demfi_avg_pool_fat': (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'demfi_absmean_map': (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'demfi_minmax_scratch_floats': (C.c_int64, []),
    'demfi_minmax_normalize': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'demfi_one_minus': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'demfi_gate_blend': (C.c_int, [C.c_void_p, C.POINTER(View), C.POINTER(View), C.POINTER(View), C.c_int, C.c_int,
                                   C.c_int, C.c_void_p]),
    'demfi_pack_planes': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    'demfi_u8_to_window': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'demfi_u8_to_planar': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'demfi_u8_ingest': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p]),
    'demfi_frame_to_u8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'demfi_eval_workspace_bytes': (C.c_int64, [C.c_int, C.c_int]),
    'demfi_eval_frame': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    'demfi_png_encode_bound': (C.c_int64, [C.c_int, C.c_int]),
    'demfi_png_encode': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                   C.POINTER(C.c_int64)]),
    'demfi_png_info': (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'demfi_png_decode': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    'demfi_conv_build': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.POINTER(ConvSrc), C.c_int, C.POINTER(ConvDst), C.c_int, C.POINTER(Conv),
                                   C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int32)]),
    'demfi_ctx_create': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(HParams), C.c_int, C.c_int,
                                   C.POINTER(C.c_void_p)]),
    'demfi_ctx_destroy': (C.c_int, [C.c_void_p]),
    'demfi_gru_sep_create': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'demfi_fgac_create': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'demfi_operator_run': (C.c_int, [C.c_void_p, C.c_void_p]),
    'demfi_load_weight': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    'demfi_ctx_workspace_bytes': (C.c_int64, [C.c_void_p]),
    'demfi_workspace_bytes': (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'demfi_ctx_bind': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'demfi_ctx_weight_region': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'demfi_ctx_buffer': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32)]),
    'demfi_ingest_u8': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'demfi_forward_trunk_body': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'demfi_forward_trunk': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'demfi_forward_t': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'demfi_forward_tb': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'demfi_forward_tb_final': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'demfi_ctx_num_ops': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    'demfi_ctx_get_op': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Op)]),
    'demfi_ctx_num_convs': (C.c_int, [C.c_void_p]),
    'demfi_ctx_conv_desc': (C.POINTER(Conv), [C.c_void_p, C.c_int]),
    'demfi_run_op': (C.c_int, [C.c_void_p, C.POINTER(Op), C.c_void_p]),
    'demfi_graph_begin': (C.c_int, [C.c_void_p]),
    'demfi_graph_end': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    'demfi_graph_launch': (C.c_int, [C.c_void_p, C.c_void_p]),
    'demfi_graph_destroy': (C.c_int, [C.c_void_p]),
}

EXPORTS = tuple(_SIGS)
_lib = None


class DemfiError(RuntimeError):
    pass


def load():
    """Load libdemfi_hip.so (once).  Fails loudly: the HIP extension IS the forward path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('demfi_amd: %s not found -- build it with demfi_amd/csrc/build.sh (or '
                           '__graft_entry__.build()); there is no CPU/PyTorch fallback for the forward path'
                           % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.demfi_abi_version() != ABI_VERSION:
        raise RuntimeError('demfi_amd: ABI version mismatch')
    _lib = lib
    return lib


def check(status, what=''):
    if status < 0:
        raise DemfiError('%s failed (%d): %s' % (what or 'libdemfi_hip call', status,
                                                 load().demfi_last_error().decode(errors='replace')))
    return status


def device_info():
    lib = load()
    name = C.create_string_buffer(64)
    ncu = C.c_int(0)
    mem = C.c_int64(0)
    st = lib.demfi_device_info(name, 64, C.byref(ncu), C.byref(mem))
    return st, name.value.decode(), ncu.value, mem.value
