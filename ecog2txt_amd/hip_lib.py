"""ctypes binding of libecog2txt_hip.so (the C ABI in include/ecog2txt_hip.h).

There is NO CPU fallback: if the shared library is missing or an entry point is
absent, importing the symbol raises.  Every wrapper raises RuntimeError with
`e2t_last_error()` when the C side returns non-zero.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# E2T_DEBUG_LIB=1 (scripts/ only): the diagnostics build of the same sources (csrc/build.sh with E2T_DEBUG=1), in which the
# kernel-variant switches and the phase-stamp buffers exist; the product library has none of them
LIB_PATH = os.path.join(_HERE, 'libecog2txt_hip_dbg.so' if os.environ.get('E2T_DEBUG_LIB') == '1' else 'libecog2txt_hip.so')

GEMM_RELU, GEMM_OUT_BF16, GEMM_ACCUMULATE, GEMM_DROPOUT, GEMM_SPLITK, GEMM_LAST_ROW_ONES, GEMM_KEEP_SLABS = 1, 2, 4, 8, 16, 32, 64
PACK_UNITS = 4            # E2T_PACK_UNITS (include/ecog2txt_hip.h): work units of a pack descriptor per workgroup


class Dropout(C.Structure):
    _fields_ = [('rate', C.c_float), ('seed', C.c_ulonglong), ('step', C.c_void_p), ('stream', C.c_uint)]


class SlabInfo(C.Structure):
    _fields_ = [('slab', C.c_void_p), ('splits', C.c_int), ('batch', C.c_int), ('stride', C.c_longlong)]


class GemmEpilogue(C.Structure):
    _fields_ = [('bias', C.c_void_p), ('relu_bwd_src', C.c_void_p), ('ld_relu_bwd_src', C.c_int),
                ('row_lens', C.c_void_p), ('rows_per_step', C.c_int), ('alpha', C.c_float), ('flags', C.c_int),
                ('drop_rate', C.c_float), ('drop_seed', C.c_ulonglong), ('drop_step', C.c_void_p),
                ('drop_stream', C.c_uint), ('drop_ld', C.c_int), ('last_col_out', C.c_void_p),
                ('splitk_ws', C.c_void_p), ('splitk_ws_bytes', C.c_size_t), ('batch', C.c_int),
                ('a_batch_stride', C.c_longlong), ('b_batch_stride', C.c_longlong), ('c_batch_stride', C.c_longlong),
                ('row_group', C.c_int), ('slabs_out', C.POINTER(SlabInfo))]


class GemmCall(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int), ('B', C.c_void_p), ('ldb', C.c_int), ('C', C.c_void_p), ('ldc', C.c_int),
                ('M', C.c_int), ('N', C.c_int), ('K', C.c_int), ('ep', C.POINTER(GemmEpilogue))]


class LstmDesc(C.Structure):
    _fields_ = [('S', C.c_int), ('B', C.c_int), ('H', C.c_int), ('ndir', C.c_int), ('ldy', C.c_int),
                ('forget_bias', C.c_float), ('drop_rate', C.c_float), ('drop_seed', C.c_ulonglong),
                ('drop_step', C.c_void_p), ('drop_stream', C.c_uint), ('rb_begin', C.c_int), ('rb_count', C.c_int)]


class PackDesc(C.Structure):
    _fields_ = [('kind', C.c_int), ('first_block', C.c_int), ('src_off', C.c_longlong), ('s0', C.c_longlong),
                ('s1', C.c_longlong), ('d0', C.c_int), ('d1', C.c_int), ('ld', C.c_int), ('pad_', C.c_int), ('dst', C.c_void_p)]


class TileImg(C.Structure):
    _fields_ = [('dst', C.c_void_p), ('kind', C.c_int), ('ld', C.c_int)]


TILE_IMG_MAX = 3          # E2T_TILE_IMG_MAX
TILE_CAST, TILE_CAST_T, TILE_FRAG_NK, TILE_FRAG_KN, TILE_FRAG4_KN = 1, 2, 3, 4, 5


class TileDesc(C.Structure):
    _fields_ = [('first_block', C.c_int), ('R', C.c_int), ('C', C.c_int), ('nimg', C.c_int), ('src_off', C.c_longlong),
                ('s0', C.c_longlong), ('gslab', C.c_void_p), ('gstride', C.c_longlong), ('gsplits', C.c_int), ('pad_', C.c_int),
                ('img', TileImg * TILE_IMG_MAX)]


class AdamHyper(C.Structure):
    _fields_ = [('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float),
                ('ema_decay', C.c_float), ('grad_scale', C.c_float), ('step_offset', C.c_int), ('skip_if_nonzero', C.c_void_p)]


_p, _i, _f, _l, _z = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_size_t

# name -> argtypes; the list mirrors include/ecog2txt_hip.h one to one
SIGNATURES = {
    'e2t_fill_u32': [_p, _z, C.c_uint32, _p],
    'e2t_gather_rows_u32': [_p, _p, _i, _i, _z, _p, _p],
    'e2t_gather_rows_blocks_u32': [_p, _p, _i, _i, _z, _i, _z, _z, _p, _p],
    'e2t_seq_lengths_f32': [_p, _i, _i, _i, _i, _p, _p, _p],
    'e2t_seq_lengths_tail_f32': [_p, _i, _i, _i, _i, _p, _p, _p],
    'e2t_seq_lengths_i32': [_p, _i, _i, _i, _i, _p, _p, _p],
    'e2t_sum_i32': [_p, _i, _p, _p],
    'e2t_sum_f32': [_p, _i, _p, _f, _p, _p],
    'e2t_sum2_f32': [_p, _p, _i, _p, _f, _f, _p, _p, _p],
    'e2t_conv_pack': [_p, _p, _i, _i, _i, _i, _p, _i, _p],
    'e2t_conv_pack_grouped': [_p, _p, _i, _i, _i, _i, _i, _p, _i, _p],
    'e2t_conv_fwd_fused': [_p, _p, _i, _i, _i, _i, _p, _i, _p, _i, _i, _p, _i, C.POINTER(GemmEpilogue), _p],
    'e2t_conv_unpack_grad': [_p, _i, _p, _i, _i, _i, _i, _p, _p],
    'e2t_conv_unpack_grad_grouped': [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p],
    'e2t_gather_rev_decim_f32': [_p, _p, _i, _i, _i, _i, _p, _p],
    'e2t_gather_rev_decim_i32': [_p, _p, _i, _i, _i, _p, _p],
    'e2t_decoder_tokens': [_p, _i, _i, _i, _p, _p, _p],
    'e2t_gemm_nt_bf16': [_p, _i, _p, _i, _p, _i, _i, _i, _i, C.POINTER(GemmEpilogue), _p],
    'e2t_gemm_tn_bf16': [_p, _i, _p, _i, _p, _i, _i, _i, _i, C.POINTER(GemmEpilogue), _p],
    'e2t_gemm_tn_group_bf16': [_i, C.POINTER(GemmCall), _p],
    'e2t_gemm_plan': [_i, _i, _i, _i, C.POINTER(GemmEpilogue), C.POINTER(_i), C.POINTER(_i)],
    'e2t_gemm_stamps': [_p, _i],
    'e2t_gemm_stamp_kinds': [C.POINTER(_i), _i],
    'e2t_transpose_bf16': [_p, _i, _i, _i, _p, _i, _p],
    'e2t_cast_pack': [_p, _l, _l, _i, _i, _p, _i, _p],
    'e2t_pack_frag': [_p, _l, _l, _i, _i, _p, _p],
    'e2t_pack_batch': [_p, _i, _i, _p, _p],
    'e2t_lstm_seq_fwd': [C.POINTER(LstmDesc), _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    'e2t_lstm_seq_fwd_persistent': [C.POINTER(LstmDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    'e2t_lstm_seq_bwd': [C.POINTER(LstmDesc), _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'e2t_lstm_seq_bwd_persistent': [C.POINTER(LstmDesc), _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    'e2t_lstm_seq_fwd_big': [C.POINTER(LstmDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    'e2t_lstm_seq_bwd_big': [C.POINTER(LstmDesc), _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    'e2t_final_state': [_p, _i, _p, _p, _i, _i, _p, _i, _p, _p],
    'e2t_embed_fwd': [_p, _i, _p, _i, _i, _i, _p, _i, C.POINTER(Dropout), _p],
    'e2t_embed_bwd': [_p, _i, _p, _i, _i, _p, _i, C.POINTER(Dropout), _p],
    'e2t_softmax_ce': [_p, _i, _i, _i, _p, _p, _i, _p, _f, _p, _p, _p, _p, _i, _p],
    'e2t_greedy_update': [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    'e2t_greedy_step': [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    'e2t_decode_init': [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    'e2t_greedy_head_small': [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p, _p, _p],
    'e2t_beam_step': [_p, _i, _i, _i, _i, _f, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'e2t_beam_reorder': [_p, _i, _p, _i, _i, _p, _p, _p, _p],
    'e2t_mse': [_p, _i, _p, _i, _i, _p, _i, _p, _f, _p, _p, _i, _p],
    'e2t_inc_step': [_p, _p, _p],
    'e2t_adam_ema_step': [_p, _p, _p, _p, _p, _z, _p, C.POINTER(AdamHyper), _p],
    'e2t_adam_pack_batch': [_p, _i, _i, _p, _p, _p, _p, _p, _p, C.POINTER(AdamHyper), _p],
    'e2t_comm_unique_id': [_p],
    'e2t_comm_init': [C.POINTER(_p), _i, _i, _p, _i],
    'e2t_comm_destroy': [_p],
    'e2t_comm_order_after': [_p, _p],
    'e2t_comm_allreduce_f32': [_p, _p, _z, _p, C.POINTER(_i)],
    'e2t_comm_allreduce_i32': [_p, _p, _z, _p, C.POINTER(_i)],
    'e2t_comm_allreduce_max_i32': [_p, _p, _z, _p, C.POINTER(_i)],
    'e2t_comm_broadcast': [_p, _p, _z, _i, _p, C.POINTER(_i)],
    'e2t_comm_wait': [_p, _i, _p],
}
COMM_ID_BYTES = 128
PLAIN = {'e2t_abi_version': ([], C.c_int), 'e2t_sizeof': ([_i], C.c_int), 'e2t_last_error': ([], C.c_char_p), 'e2t_device_cus': ([_i], C.c_int),
         'e2t_bwd_persist_kq': ([_i], C.c_int), 'e2t_lstm_big_ok': ([_i], C.c_int), 'e2t_conv_fwd_fused_ok': ([_i, _i], C.c_int), 'e2t_comm_rank': ([_p], C.c_int), 'e2t_comm_size': ([_p], C.c_int),
         'e2t_crc32c': ([_p, _z, C.c_uint32], C.c_uint32)}

_lib = None


def load():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libecog2txt_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` '
            'or ecog2txt_amd/csrc/build.sh. There is no CPU fallback for this path.' % LIB_PATH)
    # torch first: it ships its own libamdhip64 and the host side keeps tensors and streams in that runtime; loaded the other
    # way round, this library binds the ROCm install's copy and the process ends up with two HIP runtimes (the second one
    # reports "no ROCm-capable device" on the first launch -- seen with build() followed by smoke() in one process)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes, fn.restype = args, C.c_int
    _lib = lib
    return lib


class _Calls:
    """Attribute access returns a checked wrapper: lib.e2t_xxx(...) raises on failure."""

    def __getattr__(self, name):
        lib = load()
        fn = getattr(lib, name)
        if name in PLAIN:
            return fn

        def checked(*a):
            rc = fn(*a)
            if rc != 0:
                raise RuntimeError('%s failed (%d): %s' % (name, rc, lib.e2t_last_error().decode()))
        checked.__name__ = name
        setattr(self, name, checked)
        return checked


lib = _Calls()
