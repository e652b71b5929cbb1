"""ctypes binding of libgml_hip.so (the C ABI declared in include/gml.h).

There is deliberately NO fallback: if the shared library is missing or a kernel returns an error
the call raises -- the MI355X path either runs or fails loudly."""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# GML_LIB: another build of the same library (A/B and instrumented builds of tools/build_variant.py)
LIB_PATH = os.environ.get('GML_LIB') or os.path.join(_PKG, 'libgml_hip.so')

_i32, _i64, _u32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32
_p, _sz = ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/gml.h one to one
SIGNATURES = {
    'gml_version': (ctypes.c_int, []),
    'gml_error_string': (ctypes.c_char_p, [ctypes.c_int]),
    'gml_csr_workspace_bytes': (_sz, [_i64, _i64]),
    'gml_csr_from_coo': (ctypes.c_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    'gml_csr_from_sorted_coo': (ctypes.c_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    'gml_csr_link_transpose': (ctypes.c_int, [_p, _p, _i64, _p, _p, _p]),
    'gml_csr_group_record_ints': (ctypes.c_int32, [_i32]),
    'gml_csr_group_info': (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p]),
    'gml_csr_group_info2': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    'gml_batch_assemble': (ctypes.c_int, [_p, _p]),
    'gml_bn_workspace_bytes': (ctypes.c_size_t, [_i64]),
    'gml_bn_stats': (ctypes.c_int, [_p, _i64, _i64, _i32, ctypes.c_float, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    'gml_bn_apply': (ctypes.c_int, [_p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _i64, _p]),
    'gml_bn_bwd_sums': (ctypes.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    'gml_bn_bwd_apply': (ctypes.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _i64, _p]),
    'gml_gather_rows': (ctypes.c_int, [_p, _p, _p, _i64, _i32, _p]),
    'gml_gather_rows_presplit': (ctypes.c_int, [_p, _p, _p, _p, _i64, _i32, _p]),
    'gml_scatter_rows': (ctypes.c_int, [_p, _p, _p, _i64, _i32, _p]),
    'gml_spectconv_fwd_group_rows': (ctypes.c_int32, [_i32, _i32, _i32, ctypes.c_uint32]),
    'gml_spectconv_fwd_stage_edges': (ctypes.c_int32, [_i32, _i32, _i32, ctypes.c_uint32]),
    'gml_spectconv_fwd_stage_window': (ctypes.c_int32, [_i32, _i32, _i32, ctypes.c_uint32]),
    'gml_spectconv_fwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _i64,
                                         _i64, _i32, _i32, _i32, _u32, _p]),
    'gml_ml3_fwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _i64,
                                   _i64, _i32, _i32, _i32, _i32, _u32, _p]),
    'gml_spectconv_bwd_group_rows': (ctypes.c_int, [_i32, _i32, _i32, _u32]),
    'gml_spectconv_bwd_workspace_bytes': (_sz, [_i64, _i32, _i32, _i32, _i32, _i32, _u32]),
    'gml_spectconv_bwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p,
                                         _i64, _i32, _i32, _i32, _i32, _i32, _u32, _p, _sz, _p]),
    'gml_spectconv_bwd_mix_supported': (ctypes.c_int, [_i32, _i32, _i32, _i32, _u32]),
    'gml_spectconv_bwd_mix': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i32,
                                             _i64, _i32, _i32, _i32, _i32, _i32, _u32, _p, _sz, _p]),
    'gml_edge_mlp_fwd_stack': (ctypes.c_int, [_p, _i32, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    'gml_edge_mlp_fwd_stack6': (ctypes.c_int, [_p, _i32, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    'gml_edge_sym_flags': (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p, _p]),
    'gml_edge_mlp_fwd_stack6_sym': (ctypes.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    'gml_edge_mlp_bwd_sym_parts': (ctypes.c_int64, [_i64, _i32]),
    'gml_edge_mlp_bwd_sym': (ctypes.c_int, [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, ctypes.c_size_t, _p]),
    'gml_edge_mlp_fwd6': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    'gml_spectconv_bwd_mix_relu': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i32, _i32,
                                                  _i64, _i32, _i32, _i32, _i32, _i32, _u32, _p, _sz, _p]),
    'gml_spectconv_bwd_mix_relu2': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i32, _p, _i32, _i32,
                                                   _i64, _i32, _i32, _i32, _i32, _i32, _u32, _p, _sz, _p]),
    'gml_spectconv_bwd_had_parts': (ctypes.c_int, [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _u32]),
    'gml_spectconv_bwd_had': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i32,
                                             _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _u32, _p, _sz, _p, _sz, _p]),
    'gml_spectconv_fwd_epi': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _i64, _i64, _i32, _i32, _i32,
                                             ctypes.c_uint32, _i32, _p, _i32, _p]),
    'gml_spmm_fwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _p]),
    'gml_spmm_fwd_ex': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _i32, _p]),
    'gml_sddmm': (ctypes.c_int, [_p, _p, _p, _p, _i64, _p, _p, _i64, _i32, _i32, _p]),
    'gml_edge_presplit': (ctypes.c_int, [_p, _p, _i64, _i32, _p]),
    'gml_edge_mlp_fwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    'gml_edge_mlp_fwd_exact': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    'gml_edge_mlp_bwd_exact': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _sz, _p]),
    'gml_head_l1_fwd': (ctypes.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p]),
    'gml_head_l1_fwd_acc': (ctypes.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    'gml_head_l1_bwd': (ctypes.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _i64, _p, _p, _p, _p, _p]),
    'gml_head_l1_big_workspace_floats': (_sz, [_i64, _i32, _i32]),
    'gml_head_l1_big_fwd': (ctypes.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _sz, _p]),
    'gml_head_l1_big_bwd': (ctypes.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _i64, _p, _p, _p, _p,
                                           _p, _sz, _p]),
    'gml_dense_dw_slices': (_i32, [_i32]),
    'gml_dense_dw_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32]),
    'gml_dense_conv_bwd_w': (ctypes.c_int, [_p, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _sz, _p]),
    'gml_fold_many': (ctypes.c_int, [_p, _i32, _p]),
    'gml_adam_many': (ctypes.c_int, [_p, _i32, _p, _p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _p]),
    'gml_gnnml1_supported': (_i32, [_i32, _i32, _i32, _i32, _i32]),
    'gml_gnnml1_g4_cols': (_i32, [_i32, _i32, _i32, _i32]),
    'gml_gnnml1_dw_floats': (_i64, [_i32, _i32, _i32, _i32, _i32]),
    'gml_gnnml1_dw_workspace_bytes': (_sz, [_i64, _i32, _i32, _i32, _i32, _i32]),
    'gml_gnnml1_dw': (ctypes.c_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _p, _p, _sz, _p]),
    'gml_gnnml1_fwd': (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _i32, _p, _p, _i32, _p, _p, _p, _p, _i32, _i32, _i32, _p, _i64, _p]),
    'gml_gnnml1_bwd': (ctypes.c_int, [_p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i32, _p, _i32, _p, _i32, _p, _p, _p, _p, _i32, _i32, _i32,
                                      _p, _i64, _p, _i64, _p, _i64, _p]),
    'gml_edge_mlp_bwd_parts': (_i64, [_i64, _i32, _i32, _i32, _i32]),
    'gml_edge_mlp_wide_fwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    'gml_edge_mlp_wide_bwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    'gml_edge_mlp_wide_bwd_h2r': (ctypes.c_int32, [_i32]),
    'gml_edge_mlp_bwd_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'gml_edge_mlp_bwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _sz, _p]),
    'gml_node_mix_fwd': (ctypes.c_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p]),
    'gml_node_mix_bwd_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'gml_node_mix_bwd': (ctypes.c_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i32, _i32,
                                        _p, _sz, _p]),
    'gml_relu_bwd': (ctypes.c_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _p]),
    'gml_segment_sum': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _i64, _i32, _i32, _p]),
    'gml_segment_bcast': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _i64, _i32, _i32, _p]),
    'gml_segment_sum_mask': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _p, _i64, _i32, _i32, _p]),
    'gml_segment_bcast_mask': (ctypes.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _p]),
    'gml_segment_max': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _p, _i64, _i32, _p]),
    'gml_segment_max_bwd': (ctypes.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _p]),
    'gml_dense_pack': (ctypes.c_int, [_p, _p, _i64, _i32, _i32, _i32, _p]),
    'gml_dense_wimg_elems': (_sz, [_i32, _i32, _i32]),
    'gml_dense_pack_w': (ctypes.c_int, [_p, _p, _i32, _i32, _i32, _p]),
    'gml_dense_wimgt_elems': (_sz, [_i32, _i32, _i32]),
    'gml_dense_pack_wt': (ctypes.c_int, [_p, _p, _i32, _i32, _i32, _p]),
    'gml_dense_conv_bwd_x': (ctypes.c_int, [_p, _p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'gml_dense_conv_fwd': (ctypes.c_int, [_p, _p, _i64, _p, _p, _p, _i64, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'gml_dense_support_mm': (ctypes.c_int, [_p, _p, _i64, _i32, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'gml_spectral_count': (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _p, _p]),
    'gml_spectral_design': (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _i32, ctypes.c_double, _i32, ctypes.c_double,
                                           _i32, _i32, _p, _i64, _p, _p, _p, _p]),
    'gml_xty_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'gml_xty': (ctypes.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _p, _sz, _p]),
    'gml_xty_wide_supported': (_i32, [_i64, _i32, _i32]),
    'gml_xty_wide_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'gml_xty_wide': (ctypes.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _p, _sz, _p]),
    'gml_ml3_split_bwd_workspace_bytes': (_sz, [_i64, _i32, _i32, _i32]),
    'gml_ml3_split_bwd_ex': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p,
                                            _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _sz, _p]),
    'gml_ml3_split_bwd': (ctypes.c_int, [_p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _p,
                                         _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _sz, _p]),
}



class BatchDesc(ctypes.Structure):
    """gml_batch_desc of include/gml.h"""
    _fields_ = [(n, _p) for n in ('node_ptr', 'edge_ptr2', 'x', 'edge_index2', 'edge_attr2', 'es', 'tperm', 'tinv', 'rp_src', 'rp_dst', 'y')] + \
               [('G', _i64), ('E2all', _i64), ('F', _i32), ('S', _i32), ('ids', _p), ('B', _i32), ('n_pad', _i32), ('e2_pad', _i32), ('dmax', _i32)] + \
               [(n, _p) for n in ('x_out', 'ea_out', 'es_out', 'y_out', 'valid_out', 'ptr_out', 'batch_out', 'rowptr', 'col', 'perm', 'rowptr_t',
                                  'col_t', 'pos_t')] + [('ldx_out', _i32), ('nblk_main', _i32), ('ginfo128', _p), ('ginfo_t128', _p)]


GML_OK, GML_E_BADARG, GML_E_UNSUPPORTED, GML_E_WORKSPACE = 0, -1, -2, -3
GML_RELU, GML_ACCUM, GML_F32_MFMA, GML_GROUPS128, GML_GROUPS64R, GML_DMA_RING, GML_FWD_CHUNKED, GML_DVAL_ACCUM = 1, 2, 4, 8, 16, 32, 64, 128
GML_POOL_SKIP_LAST = 2
GML_FWD_ONEWIN = 256
GML_F16X3 = 1024                # gml_spectconv_fwd / gml_ml3_fwd, ring kernel: f16 (hi, lo) pieces under power-of-two scales
GML_NO_FOLD = 512               # gml_spectconv_bwd*: leave the dW partials in ws (gml_fold_many)
GML_FOLD_MAX_JOBS = 16


class AdamJob(ctypes.Structure):
    """gml_adam_job of include/gml.h"""
    _fields_ = [('p', _p), ('g', _p), ('m', _p), ('v', _p), ('n', _i64)]
GML_ADAM_MAX_JOBS = 64


class FoldJob(ctypes.Structure):
    """gml_fold_job of include/gml.h"""
    _fields_ = [('partial', _p), ('nparts', _i64), ('n', _i64), ('dst', _p * 5), ('ndst', _i64 * 5)]
GML_GROUPS64_RANKED = 1064      # group kind of gml_csr_group_info: 64-row groups with rank bytes

_lib = None


class GmlError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                '%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(or python gnn_matlang_amd/_build.py). There is no CPU fallback.' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().gml_error_string(int(rc))
        raise GmlError('libgml_hip: %s (code %d)' % (msg.decode() if msg else '?', rc))


def call(name, *args):
    check(getattr(lib(), name)(*args))
