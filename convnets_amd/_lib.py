"""ctypes binding of libcnk.so -- the C-ABI of include/cnk.h (built by convnets_amd/build.py).

There is NO fallback: if the shared library is missing or a call fails, this raises.  The
oracle (oracle/) is never imported from here.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libcnk.so')

CNK_F32, CNK_BF16 = 0, 1
PASS_FWD, PASS_DGRAD, PASS_WGRAD = 0, 1, 2
# kernel-level scopes (csrc/common.h CNK_KF_*): one record per launch of that kernel family
KF = dict(igemm_gather_kernel=1 << 16, igemm_halo3x3_kernel=1 << 17, wgrad_mfma_kernel=1 << 18,
          bn_apply_kernel=1 << 19, bn_bwd_apply_kernel=1 << 20)
CLS = dict(conv3x3_fwd=1, conv3x3_dgrad=2, conv3x3_wgrad=4, conv1x1_fwd=8, conv1x1_dgrad=16,
           conv1x1_wgrad=32, bn_fwd=64, bn_bwd=128, other=256)


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ('N', 'H', 'W', 'Cin', 'Cout', 'kh', 'kw', 'stride_h', 'stride_w', 'pad_h', 'pad_w',
                 'dtype', 'groups', 'dilation')]


class BnBwdFuse(ctypes.Structure):
    """cnk_bn_bwd_fuse (include/cnk.h): BN backward folded into the conv dgrad epilogue."""
    _fields_ = [(n, ctypes.c_void_p) for n in ('x', 'mean', 'invstd', 'gamma', 'beta', 'relu_mask', 'partial')] + \
               [('relu', ctypes.c_int32)]


class PackJob(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('out', ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ('Cout', 'Cin', 'KH', 'KW', 'ntaps', 'transpose', 'dtype', 'groups')] + \
               [('kh', ctypes.c_int8 * 32), ('kw', ctypes.c_int8 * 32), ('nblocks', ctypes.c_int32), ('pad_', ctypes.c_int32)]


_vp, _i32, _i64, _f32, _sz = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float,
                              ctypes.c_size_t)
_dp = ctypes.POINTER(ConvDesc)

class WgradPending(ctypes.Structure):
    """cnk_wgrad_pending (include/cnk.h): the slab sum a chained weight gradient left to the next launch of its stream."""
    _fields_ = [('slab', ctypes.c_void_p), ('dw', ctypes.c_void_p), ('slab_stride', ctypes.c_uint64)] + \
               [(n, ctypes.c_int32) for n in ('splits', 'Cout', 'Cin', 'ntaps', 'accumulate', 'lanes')]


# name -> (restype, argtypes); must list every symbol declared in include/cnk.h
SIGNATURES = {
    'cnk_version': (ctypes.c_int, []),
    'cnk_last_error': (ctypes.c_char_p, []),
    'cnk_stream_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    'cnk_stream_destroy': (ctypes.c_int, [_vp]),
    'cnk_graph_anchor': (ctypes.c_int, [_vp]),
    'cnk_conv2d_out_hw': (ctypes.c_int, [_dp, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    'cnk_conv2d_packed_bytes': (_sz, [_dp, ctypes.c_int]),
    'cnk_conv2d_workspace_bytes': (_sz, [_dp, ctypes.c_int]),
    'cnk_conv2d_pack_weights': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp]),
    'cnk_conv2d_pack_jobs': (ctypes.c_int, [_dp, _vp, _vp, _vp, ctypes.POINTER(PackJob), _i32]),
    'cnk_pack_batch': (ctypes.c_int, [_vp, _vp, _i32, _vp]),
    'cnk_conv2d_fwd_bn_partial_slices': (_i64, [_dp]),
    'cnk_conv2d_fwd': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'cnk_conv2d_fwd_act_supported': (ctypes.c_int, [_dp]),
    'cnk_conv2d_fwd_act': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _vp]),
    'cnk_conv2d_dgrad_accepts_mask': (ctypes.c_int, [_dp]),
    'cnk_conv2d_dgrad_bn_slices': (_i64, [_dp]),
    'cnk_conv2d_dgrad': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(BnBwdFuse), _vp, _sz, _vp]),
    'cnk_conv2d_wgrad': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _sz, _vp]),
    'cnk_conv2d_wgrad_min_workspace_bytes': (_sz, [_dp, ctypes.c_int]),
    'cnk_conv2d_wgrad_chain_supported': (ctypes.c_int, [_dp]),
    'cnk_conv2d_wgrad_chain': (ctypes.c_int, [_dp, _vp, _vp, _vp, ctypes.c_int, _vp, _sz, ctypes.POINTER(WgradPending),
                                              ctypes.POINTER(WgradPending), _vp]),
    'cnk_wgrad_reduce_pending': (ctypes.c_int, [ctypes.POINTER(WgradPending), _vp]),
    'cnk_conv2d_wgrad_bnfold_chain': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _sz,
                                                     ctypes.POINTER(WgradPending), ctypes.POINTER(WgradPending), _vp]),
    'cnk_conv2d_bnfold_supported': (ctypes.c_int, [_dp]),
    'cnk_conv2d_bwd_bnfold_supported': (ctypes.c_int, [_dp]),
    'cnk_conv2d_bwd_bnfold_workspace_bytes': (_sz, [_dp]),
    'cnk_conv2d_bwd_bnfold_bn_slices': (ctypes.c_int64, [_dp]),
    'cnk_conv2d_bwd_bnfold': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(BnBwdFuse), _vp, _vp, _vp, _vp, _vp, ctypes.c_int,
                              _vp, _sz, _vp]),
    'cnk_conv2d_bnfold_image_bytes': (_sz, [_dp]),
    'cnk_bn_bwd_coef': (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _i64, _i32, ctypes.c_int, _vp,
                                       _sz, _vp]),
    'cnk_conv2d_bnfold_prepare': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'cnk_conv2d_dgrad_bnfold': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(BnBwdFuse), _vp]),
    'cnk_conv2d_wgrad_bnfold': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _sz, _vp]),
    'cnk_bn_workspace_bytes': (_sz, [_i64, _i32]),
    'cnk_bn_mask_bytes': (_sz, [_i64, _i32, ctypes.c_int]),
    'cnk_bn_fwd_train': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32,
                                        _f32, _f32, ctypes.c_int, ctypes.c_int, _vp, _sz, _vp]),
    'cnk_bn_fwd_train_res_affine': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32,
                                                   _f32, _f32, ctypes.c_int, ctypes.c_int, _vp, _sz, _vp]),
    'cnk_bn_fwd_eval': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32,
                                       ctypes.c_int, ctypes.c_int, _vp]),
    'cnk_bn_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int,
                                  _vp, _i64, _i64, _i32, ctypes.c_int, ctypes.c_int, _vp, _sz, _vp]),
    'cnk_avgpool2d_fwd': (ctypes.c_int, [_vp, _vp] + [_i32] * 10 + [ctypes.c_int, _vp]),
    'cnk_avgpool2d_bwd': (ctypes.c_int, [_vp, _vp] + [_i32] * 10 + [ctypes.c_int, _vp]),
    'cnk_maxpool2d_fwd': (ctypes.c_int, [_vp, _vp, _vp] + [_i32] * 10 + [ctypes.c_int, _vp]),
    'cnk_maxpool2d_bwd': (ctypes.c_int, [_vp, _vp, _vp] + [_i32] * 10 + [ctypes.c_int, _vp]),
    'cnk_maxpool2d_fwd_bn': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp] + [_i32] * 10 + [ctypes.c_int, _vp]),
    'cnk_bn_bwd_maxpool': (ctypes.c_int, [_vp] * 10 + [ctypes.c_int] + [_i32] * 10 + [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    'cnk_maxpool2d_fwd_bn_keep': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp] + [_i32] * 10 + [ctypes.c_int, _vp]),
    'cnk_bn_bwd_maxpool_kept': (ctypes.c_int, [_vp] * 11 + [ctypes.c_int] + [_i32] * 10 + [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    'cnk_relu_fwd': (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, _vp]),
    'cnk_relu_bwd': (ctypes.c_int, [_vp, _vp, _vp, _i64, ctypes.c_int, _vp]),
    'cnk_add_relu_fwd': (ctypes.c_int, [_vp, _vp, _vp, _i64, ctypes.c_int, _vp]),
    'cnk_gap_fwd': (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, ctypes.c_int, _vp]),
    'cnk_gap_bwd': (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, ctypes.c_int, _vp]),
    'cnk_nchw_to_nhwc': (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, ctypes.c_int, _vp]),
    'cnk_dwconv2d_supported': (ctypes.c_int, [_dp]),
    'cnk_dwconv2d_workspace_bytes': (_sz, [_dp]),
    'cnk_dwconv2d_fwd': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, _vp]),
    'cnk_dwconv2d_dgrad': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp]),
    'cnk_dwconv2d_wgrad': (ctypes.c_int, [_dp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _sz, _vp]),
    'cnk_grouped_expand': (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'cnk_grouped_extract': (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'cnk_channel_shuffle': (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, ctypes.c_int, _vp]),
    'cnk_linear_workspace_bytes': (_sz, [ctypes.c_int, _i32, _i32, _i32]),
    'cnk_linear_fwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    'cnk_linear_dgrad': (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    'cnk_linear_wgrad': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, _i32, _i32, _i32, _vp, _sz, _vp]),
    'cnk_softmax_ce_sum': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    'cnk_dropout_fwd': (ctypes.c_int, [_vp, _vp, _vp, _i64, _f32, ctypes.c_uint64, ctypes.c_uint64, _vp, ctypes.c_int, _vp]),
    'cnk_dropout_bwd': (ctypes.c_int, [_vp, _vp, _vp, _i64, _f32, ctypes.c_int, _vp]),
    'cnk_sigmoid_fwd': (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    'cnk_sigmoid_bwd': (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp]),
    'cnk_channel_scale_fwd': (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, ctypes.c_int, _vp]),
    'cnk_channel_scale_workspace_bytes': (ctypes.c_size_t, [_i32, _i32, _i32, ctypes.c_int]),
    'cnk_channel_scale_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    'cnk_sk_workspace_bytes': (ctypes.c_size_t, [_i32, _i32, _i32, _i32]),
    'cnk_sk_pool_fwd': (ctypes.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    'cnk_sk_select_fwd': (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, ctypes.c_int, _vp]),
    'cnk_sk_select_bwd_logits': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, ctypes.c_int, _vp,
                                                ctypes.c_size_t, _vp]),
    'cnk_sk_maps_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, ctypes.c_int, _vp]),
    'cnk_input_pipeline': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [_i32] * 7 + [ctypes.c_int, _vp]),
    'cnk_input_pipeline_affine': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp] + [_i32] * 7 + [ctypes.c_int, _vp]),
    'cnk_affine_fixed_coeffs': (ctypes.c_int, [ctypes.c_double] * 6 + [_i32, _i32, ctypes.POINTER(ctypes.c_int32)]),
    'cnk_grad_clip_workspace_bytes': (_sz, []),
    'cnk_grad_clip_norm': (ctypes.c_int, [_vp, _i64, _f32, _vp, _vp, _sz, _vp]),
    'cnk_grad_clip_value': (ctypes.c_int, [_vp, _i64, _f32, _vp]),
    'cnk_adam_step': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32,
                                     _vp]),
    'cnk_adam_bias_corrections': (None, [_f32, _f32, _i32, ctypes.POINTER(ctypes.c_float)]),
    'cnk_adam_step_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp, _vp]),
    'cnk_prof_begin': (ctypes.c_int, [_i32]),
    'cnk_prof_collect': (ctypes.c_int, [_i32, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double)]),
    'cnk_prof_end': (ctypes.c_int, []),
}

_lib = None


def load():
    """Loads libcnk.so and types every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'convnets_amd: %s is missing -- the HIP library is the product; there is no CPU '
                'fallback. Build it with `python -m convnets_amd.build` (needs hipcc).' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class switches:
    """Test / A-B hook: the library resolves its CNK_* switches once, when it is loaded (csrc/common.h CnkSwitches).
    `with switches(CNK_PP_CFG='2'): ...` sets the variables, makes the library re-resolve them
    (cnk_debug_reload_switches, exported but not part of include/cnk.h) and restores both on exit."""

    def __init__(self, **env):
        self.env = env
        self.prev = {}

    def __enter__(self):
        for k, v in self.env.items():
            self.prev[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        reload_switches()
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        reload_switches()
        return False


_switch_epoch = [0]


def switch_epoch():
    """Moves whenever the library re-resolved its switches: host-side caches of descriptor-derived facts key on it."""
    return _switch_epoch[0]


def reload_switches():
    _switch_epoch[0] += 1
    lib = load()
    fn = lib.cnk_debug_reload_switches
    fn.restype, fn.argtypes = None, []
    fn()


class CnkError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise CnkError('libcnk error %d: %s' % (rc, load().cnk_last_error().decode()))
