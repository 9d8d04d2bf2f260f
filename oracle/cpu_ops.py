"""ORACLE (test infrastructure, NOT product code): ctypes/numpy front-end of oracle/cpu_ops.c.

NCHW fp32 in and out, like the reference's tensors (basemodel.py:646).  Built by
`make -C oracle` (gcc) into oracle/_build/liboracle.so; __graft_entry__.build() runs that recipe.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i64 = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    src = os.path.join(_HERE, 'cpu_ops.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_f)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _out_hw(H, W, k, s, p):
    return (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def conv2d_fwd(x, w, bias=None, stride=1, padding=0):
    x, w = _c(x), _c(w)
    s, p = _pair(stride), _pair(padding)
    N, Ci, H, W = x.shape
    Co, _, kh, kw = w.shape
    Ho, Wo = _out_hw(H, W, (kh, kw), s, p)
    y = np.empty((N, Co, Ho, Wo), np.float32)
    b = None if bias is None else _c(bias)
    lib().orc_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), N, Ci, H, W, Co, kh, kw, s[0], s[1], p[0], p[1])
    return y


def conv2d_dgrad(dy, w, in_hw, stride=1, padding=0):
    dy, w = _c(dy), _c(w)
    s, p = _pair(stride), _pair(padding)
    N = dy.shape[0]
    Co, Ci, kh, kw = w.shape
    H, W = in_hw
    dx = np.empty((N, Ci, H, W), np.float32)
    lib().orc_conv2d_dgrad(_p(dy), _p(w), _p(dx), N, Ci, H, W, Co, kh, kw, s[0], s[1], p[0], p[1])
    return dx


def conv2d_wgrad(dy, x, ksize, stride=1, padding=0, with_bias=False):
    dy, x = _c(dy), _c(x)
    s, p, k = _pair(stride), _pair(padding), _pair(ksize)
    N, Ci, H, W = x.shape
    Co = dy.shape[1]
    dw = np.empty((Co, Ci, k[0], k[1]), np.float32)
    db = np.empty((Co,), np.float32) if with_bias else None
    lib().orc_conv2d_wgrad(_p(dy), _p(x), _p(dw), _p(db), N, Ci, H, W, Co, k[0], k[1], s[0], s[1],
                           p[0], p[1])
    return (dw, db) if with_bias else dw


def bn_fwd_train(x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """Returns y, save_mean, save_invstd; updates running_* in place when given."""
    x = _c(x)
    N, C, H, W = x.shape
    y = np.empty_like(x)
    sm = np.empty((C,), np.float32)
    si = np.empty((C,), np.float32)
    lib().orc_bn_fwd_train(_p(x), _p(_c(gamma)), _p(_c(beta)), _p(y), _p(running_mean),
                           _p(running_var), _p(sm), _p(si), N, C, H * W,
                           ctypes.c_float(eps), ctypes.c_float(momentum))
    return y, sm, si


def bn_fwd_eval(x, gamma, beta, rmean, rvar, eps=1e-5):
    x = _c(x)
    N, C, H, W = x.shape
    y = np.empty_like(x)
    lib().orc_bn_fwd_eval(_p(x), _p(_c(gamma)), _p(_c(beta)), _p(_c(rmean)), _p(_c(rvar)), _p(y),
                          N, C, H * W, ctypes.c_float(eps))
    return y


def bn_bwd(dy, x, gamma, save_mean, save_invstd):
    dy, x = _c(dy), _c(x)
    N, C, H, W = x.shape
    dx = np.empty_like(x)
    dg = np.empty((C,), np.float32)
    db = np.empty((C,), np.float32)
    lib().orc_bn_bwd(_p(dy), _p(x), _p(_c(gamma)), _p(_c(save_mean)), _p(_c(save_invstd)), _p(dx),
                     _p(dg), _p(db), N, C, H * W)
    return dx, dg, db


def relu_fwd(x):
    x = _c(x)
    y = np.empty_like(x)
    lib().orc_relu_fwd(_p(x), _p(y), ctypes.c_size_t(x.size))
    return y


def relu_bwd(dy, y):
    dy, y = _c(dy), _c(y)
    dx = np.empty_like(dy)
    lib().orc_relu_bwd(_p(dy), _p(y), _p(dx), ctypes.c_size_t(dy.size))
    return dx


def add_relu_fwd(a, b):
    a, b = _c(a), _c(b)
    y = np.empty_like(a)
    lib().orc_add_relu_fwd(_p(a), _p(b), _p(y), ctypes.c_size_t(a.size))
    return y


def maxpool_fwd(x, ksize, stride, padding=0):
    x = _c(x)
    k, s, p = _pair(ksize), _pair(stride), _pair(padding)
    N, C, H, W = x.shape
    Ho, Wo = _out_hw(H, W, k, s, p)
    y = np.empty((N, C, Ho, Wo), np.float32)
    idx = np.empty((N, C, Ho, Wo), np.int64)
    lib().orc_maxpool_fwd(_p(x), _p(y), idx.ctypes.data_as(_i64), N, C, H, W, k[0], k[1], s[0],
                          s[1], p[0], p[1])
    return y, idx


def maxpool_bwd(dy, idx, in_hw):
    dy = _c(dy)
    N, C, Ho, Wo = dy.shape
    H, W = in_hw
    dx = np.empty((N, C, H, W), np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    lib().orc_maxpool_bwd(_p(dy), idx.ctypes.data_as(_i64), _p(dx), N, C, H, W, Ho, Wo)
    return dx


def gap_fwd(x):
    x = _c(x)
    N, C, H, W = x.shape
    y = np.empty((N, C, 1, 1), np.float32)
    lib().orc_gap_fwd(_p(x), _p(y), N, C, H * W)
    return y


def gap_bwd(dy, in_hw):
    dy = _c(dy)
    N, C = dy.shape[:2]
    H, W = in_hw
    dx = np.empty((N, C, H, W), np.float32)
    lib().orc_gap_bwd(_p(dy), _p(dx), N, C, H * W)
    return dx
