"""Thin Python wrappers over the C-ABI entry points of the denoiser kernels.

Tensors are NHWC fp32 CUDA tensors.  A "lazy source" is a tuple (y, coef, relu6): the raw conv
output, the [scale|shift|mean|invstd|mean_lo] coefficient vector of the BatchNorm that follows it (or
None) and whether a ReLU6 follows.  A "lazy gradient" is (dz, y, coef3) with
coef3 = [s|k1_hi|k1_lo|a2|mean_hi|mean_lo] or None: dy = s*(dz - k1) + a2*(y - mean).
"""
import ctypes

import torch

from . import _lib
from ._lib import ptr

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

_lib.register({
    "noise_fold_jobs": [_P, _I, _I, _P],
    "noise_dw3x3_fwd": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "noise_dw3x3_bwd": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "noise_first_conv_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P],
    "noise_first_conv_wgrad": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P],
    "noise_last_convT_fwd": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "noise_last_convT_bwd": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "noise_bn_finalize_fwd": [_P, _I, _I, _L, _P, _P, _P, _P, _P, _F, _F, _P, _P],
    "noise_bn_coef_eval": [_I, _P, _P, _P, _P, _F, _P, _P],
    "noise_bn_bwd": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _P],
    "noise_pw_fwd": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _L, _I, _I, _P],
    "noise_pw_fwd_virt_supported": [_I, _I],
    "noise_pw_fwd_virt": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "noise_pw_bwd_virt": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "noise_pw_dgrad": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "noise_pw_wgrad": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _L, _I, _I, _P],
    "noise_pw_bwd": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _P],
    "noise_bn_bwd_finalize": [_P, _I, _I, _P, _P, _P, _P, _P, _L, _I, _P],
    "noise_crop_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "noise_crop_nhwc_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
})
_lib.lib.noise_conv_max_blocks.restype = ctypes.c_int
_lib.lib.noise_pw_wgrad_max_blocks.restype = ctypes.c_int
CONV_MAX_BLOCKS = _lib.lib.noise_conv_max_blocks()
PW_WGRAD_MAX_BLOCKS = _lib.lib.noise_pw_wgrad_max_blocks()
_lib.lib.noise_pw_bwd_max_rows.restype = ctypes.c_int
_lib.lib.noise_pw_bwd_max_rows.argtypes = [ctypes.c_int, ctypes.c_int]


def pw_bwd_max_rows(Cout, Cin):
    return _lib.lib.noise_pw_bwd_max_rows(Cout, Cin)

class FoldJob(ctypes.Structure):
    """noise_fold_job of include/noise_hip.h"""
    _fields_ = [("slab", ctypes.c_void_p), ("out", ctypes.c_void_p), ("rows", ctypes.c_int), ("ncols", ctypes.c_int),
                ("row_stride", ctypes.c_int), ("tr_k", ctypes.c_int), ("tr_C", ctypes.c_int)]


NONE = (None, None, 0)


def _src(s):
    y, coef, relu6 = s if s is not None else NONE
    return ptr(y), ptr(coef), int(bool(relu6))


def _nsrc(b):
    return 2 if (b is not None and b[0] is not None) else 1


def conv_out(n, s, transposed):
    return (n - 1) * s + 1 if transposed else (n - 1) // s + 1


def crop_nhwc(src, dst, oy, ox):
    """dst (B,Hd,Wd,C) = src (B,Hs,Ws,C)[:, oy:oy+Hd, ox:ox+Wd]."""
    B, Hs, Ws, C = src.shape
    _, Hd, Wd, _ = dst.shape
    _lib.call("noise_crop_nhwc", ptr(src), ptr(dst), B, Hs, Ws, C, oy, ox, Hd, Wd, _lib.stream(),
              nbytes=8 * dst.numel())


def crop_nhwc_bwd(g_dst, g_src, oy, ox):
    """g_src (B,Hs,Ws,C) = g_dst (B,Hd,Wd,C) placed at rows oy.., columns ox.., zero elsewhere."""
    B, Hs, Ws, C = g_src.shape
    _, Hd, Wd, _ = g_dst.shape
    _lib.call("noise_crop_nhwc_bwd", ptr(g_dst), ptr(g_src), B, Hs, Ws, C, oy, ox, Hd, Wd, _lib.stream(),
              nbytes=4 * (g_dst.numel() + g_src.numel()))


def dw3x3_fwd(a, b, w, out, slab, transposed, B, Hi, Wi, C, stride):
    """Returns the number of slab rows written (0 when slab is None).  out None (stride-2 transposed layers only):
    statistics pass, the output is not written."""
    rows = ctypes.c_int(0)
    _lib.call("noise_dw3x3_fwd", *_src(a), *_src(b), ptr(w), ptr(out), ptr(slab), ctypes.byref(rows),
              int(transposed), B, Hi, Wi, C, stride[0], stride[1], _lib.stream(),
              nbytes=4 * C * (B * Hi * Wi * _nsrc(b) + (out.numel() // C if out is not None else 0)))
    return rows.value


def _epi(epi):
    coef, relu6 = epi if epi is not None else (None, 0)
    return ptr(coef), int(bool(relu6))


def dw3x3_bwd(grad, a, b, w, g_in, dw, scratch, transposed, B, Hi, Wi, C, stride, g_other=None, epi=None):
    """epi = (producer coef, producer relu6) or None.  Returns the slab row count; with epi the producer's
    BatchNorm-backward sums are in scratch columns [9C, 11C) of rows that are 11*C floats long."""
    dz, y, coef3 = grad
    rows = ctypes.c_int(0)
    # stride-2 transposed layers recompute their raw output from the (4x smaller) input instead of reading it back, and so
    # do the one-row kernels (one-row tensor, one source, stride 1 or a column stride of 2 with another row stride)
    one_row = Hi == 1 and b is None and (tuple(stride) == (1, 1) or (stride[1] == 2 and stride[0] != 2))
    reads_y = coef3 is not None and not (transposed and tuple(stride) == (2, 2)) and not one_row
    _lib.call("noise_dw3x3_bwd", ptr(dz), ptr(y), ptr(coef3), *_src(a), *_src(b), ptr(w), ptr(g_in), ptr(dw),
              ptr(scratch), ptr(g_other), *_epi(epi), ctypes.byref(rows), int(transposed), B, Hi, Wi, C, stride[0],
              stride[1], _lib.stream(),
              nbytes=4 * C * (B * Hi * Wi * (_nsrc(b) + 1 + (g_other is not None))) + dz.numel() * 4 * (2 if reads_y else 1))
    return rows.value


def first_conv_fwd(logmag, absmax, w, out, slab, B, H, W, hp, wp, pad_value, C, stride):
    rows = ctypes.c_int(0)
    _lib.call("noise_first_conv_fwd", ptr(logmag), ptr(absmax), ptr(w), ptr(out), ptr(slab), ctypes.byref(rows),
              B, H, W, hp, wp, float(pad_value), C, stride[0], stride[1], _lib.stream(),
              nbytes=4 * (B * H * W + out.numel()))
    return rows.value


def first_conv_wgrad(grad, logmag, absmax, dw, scratch, B, H, W, hp, wp, pad_value, C, stride, w=None):
    """w: the layer's weights; with them the stride-2 kernel rebuilds the raw output instead of reading grad's y."""
    dz, y, coef3 = grad
    reads_y = coef3 is not None and not (w is not None and tuple(stride) == (2, 2))
    _lib.call("noise_first_conv_wgrad", ptr(dz), ptr(y), ptr(coef3), ptr(logmag), ptr(absmax), ptr(w), ptr(dw),
              ptr(scratch), B, H, W, hp, wp, float(pad_value), C, stride[0], stride[1], _lib.stream(),
              nbytes=4 * (B * H * W + dz.numel() * (2 if reads_y else 1)))


def last_convT_fwd(a, b, w, bias, logmag, absmax, enh, filt, B, Hi, Wi, C, stride, H, W):
    _lib.call("noise_last_convT_fwd", *_src(a), *_src(b), ptr(w), ptr(bias), ptr(logmag), ptr(absmax), ptr(enh),
              ptr(filt), B, Hi, Wi, C, stride[0], stride[1], H, W, _lib.stream(),
              nbytes=4 * (B * Hi * Wi * C * _nsrc(b) + 2 * B * H * W))


def last_convT_bwd(d_enh, enh, absmax, a, b, w, g_in, dw, dbias, scratch, B, Hi, Wi, C, stride, H, W, g_other=None,
                   epi=None):
    """Scratch rows are 12*C floats; with epi the producer sums are in columns [10C, 12C).  Returns the row count."""
    rows = ctypes.c_int(0)
    _lib.call("noise_last_convT_bwd", ptr(d_enh), ptr(enh), ptr(absmax), *_src(a), *_src(b), ptr(w), ptr(g_in),
              ptr(dw), ptr(dbias), ptr(scratch), ptr(g_other), *_epi(epi), ctypes.byref(rows), B, Hi, Wi, C, stride[0],
              stride[1], H, W, _lib.stream(),
              nbytes=4 * (B * Hi * Wi * C * (_nsrc(b) + 1) + 2 * B * H * W))
    return rows.value


def bn_finalize_fwd(slab, rows, C, count, gamma, beta, running_mean, running_var, nbt, coef, momentum=0.1, eps=1e-5):
    _lib.call("noise_bn_finalize_fwd", ptr(slab), rows, C, count, ptr(gamma), ptr(beta), ptr(running_mean),
              ptr(running_var), ptr(nbt), float(momentum), float(eps), ptr(coef), _lib.stream())


def bn_coef_eval(C, gamma, beta, running_mean, running_var, coef, eps=1e-5):
    _lib.call("noise_bn_coef_eval", C, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps),
              ptr(coef), _lib.stream())


def bn_bwd(g1, g2, y, coef, relu6, gamma, dz, dgamma, dbeta, coef3, scratch, npix, C):
    _lib.call("noise_bn_bwd", ptr(g1), ptr(g2), ptr(y), ptr(coef), int(bool(relu6)), ptr(gamma), ptr(dz), ptr(dgamma),
              ptr(dbeta), ptr(coef3), ptr(scratch), npix, C, _lib.stream(),
              nbytes=4 * npix * C * (2 + (g2 is not None) + (dz is not None)))


def pw_fwd(a, b, w, out, slab, npix, K, N):
    rows = ctypes.c_int(0)
    _lib.call("noise_pw_fwd", *_src(a), *_src(b), ptr(w), ptr(out), ptr(slab), ctypes.byref(rows), npix, K, N,
              _lib.stream(), nbytes=4 * npix * (K * _nsrc(b) + N))
    return rows.value


def pw_fwd_virt_supported(K, N):
    return bool(_lib.lib.noise_pw_fwd_virt_supported(K, N))


def pw_fwd_virt(src, wdw, dw_coef, dw_relu6, w, out, slab, B, Hi, Wi, K, N):
    """Projection of a transposed block straight from the depthwise layer's input `src` (lazy source): the depthwise
    output (B, 2Hi-1, 2Wi-1, K) is rebuilt on the fly and never touches HBM."""
    rows = ctypes.c_int(0)
    npix = B * (2 * Hi - 1) * (2 * Wi - 1)
    _lib.call("noise_pw_fwd_virt", *_src(src), ptr(wdw), ptr(dw_coef), int(dw_relu6), ptr(w), ptr(out), ptr(slab),
              ctypes.byref(rows), B, Hi, Wi, K, N, _lib.stream(), nbytes=4 * (B * Hi * Wi * K + npix * N))
    return rows.value


def pw_bwd_virt(grad, src, wdw, dw_coef, dw_relu6, w, g_in, scratch, stats_slab, B, Hi, Wi, K, N):
    """Backward of pw_fwd_virt: input gradient (B, 2Hi-1, 2Wi-1, K), weight-gradient slab rows in `scratch`, the
    depthwise BatchNorm's backward sums in `stats_slab`.  Returns the number of rows of both."""
    dz, y, coef3 = grad
    rows = ctypes.c_int(0)
    npix = B * (2 * Hi - 1) * (2 * Wi - 1)
    _lib.call("noise_pw_bwd_virt", ptr(dz), ptr(y), ptr(coef3), *_src(src), ptr(wdw), ptr(dw_coef), int(dw_relu6),
              ptr(w), ptr(g_in), ptr(scratch), ptr(stats_slab), ctypes.byref(rows), B, Hi, Wi, K, N, _lib.stream(),
              nbytes=4 * (npix * (N * (2 if coef3 is not None else 1) + K) + B * Hi * Wi * K))
    return rows.value


def pw_dgrad(grad, w, add, g_in, npix, Cout, Cin):
    dz, y, coef3 = grad
    _lib.call("noise_pw_dgrad", ptr(dz), ptr(y), ptr(coef3), ptr(w), ptr(add), ptr(g_in), npix, Cout, Cin,
              _lib.stream(), nbytes=4 * npix * (Cout * (2 if coef3 is not None else 1) + Cin * (2 if add is not None else 1)))


def pw_wgrad(grad, a, b, dw, scratch, npix, Cout, Cin):
    dz, y, coef3 = grad
    _lib.call("noise_pw_wgrad", ptr(dz), ptr(y), ptr(coef3), *_src(a), *_src(b), ptr(dw), ptr(scratch), npix, Cout,
              Cin, _lib.stream(), nbytes=4 * npix * (Cout * (2 if coef3 is not None else 1) + Cin * _nsrc(b)))


def pw_bwd(grad, a, b, w, g_in, dw, scratch, npix, Cout, Cin, g_other=None, epi=None, stats_slab=None):
    """Fused dgrad + wgrad (+ producer BatchNorm-backward sums into stats_slab when epi is given).  Returns rows."""
    dz, y, coef3 = grad
    rows = ctypes.c_int(0)
    # expanding layers recompute their raw output from the staged input tile instead of reading it back
    reads_y = coef3 is not None and bool(_lib.lib.noise_pw_bwd_reads_y(Cout, Cin))
    _lib.call("noise_pw_bwd", ptr(dz), ptr(y), ptr(coef3), *_src(a), *_src(b), ptr(w), ptr(g_in), ptr(dw), ptr(scratch),
              ptr(g_other), *_epi(epi), ptr(stats_slab), ctypes.byref(rows), npix, Cout, Cin, _lib.stream(),
              nbytes=4 * npix * (Cout * (2 if reads_y else 1) + Cin * (_nsrc(b) + 1 + (g_other is not None))))
    return rows.value


def bn_bwd_finalize(slab, rows, row_stride, coef, gamma, dgamma, dbeta, coef3, npix, C):
    _lib.call("noise_bn_bwd_finalize", ptr(slab), rows, row_stride, ptr(coef), ptr(gamma), ptr(dgamma), ptr(dbeta),
              ptr(coef3), npix, C, _lib.stream())
