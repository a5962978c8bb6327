"""Training step of the configurations the tuned schedule does not cover (the waveform-domain U-Net of
train/configs/mobilenetv2.yaml): `Base._shared_step` + backward + AdamW (train/models.py:641-685, :722-726) on the
shape-generic kernels.

The forward is `generic.GenericForward` with a tape: every layer keeps what its backward needs (raw convolution outputs and
their BatchNorm coefficients, the LayerNorm inputs, q/k/v, ...).  The backward walks the tape in reverse.  Gradients of
materialised tensors (block outputs, token matrices) are plain tensors, summed where a tensor has several consumers
(next block, residual connection, decoder skip); the gradient of a convolution's raw output stays in the lazy form
(dz, y, coef3) that `noise_bn_bwd` leaves and the convolution backward kernels evaluate on load.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ptr
from .generic import GenericForward, _taps3x3, finish, normalize

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_lib.register({
    "noise_gen_slab_reduce": [_P, _I, _L, _P, _P],
    "noise_gen_pw_dgrad": [_P, _P, _P, _P, _I, _P, _L, _I, _I, _P],
    "noise_gen_pw_wgrad": [_P, _P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _P],
    "noise_gen_colsum": [_P, _I, _L, _I, _P, _P, _P],
    "noise_gen_relu_bwd": [_P, _P, _P, _L, _P],
    "noise_gen_add_cropped": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "noise_gen_ln_bwd": [_P, _P, _P, _F, _P, _P, _P, _P, _P, _L, _I, _P],
    "noise_gen_attn_bwd": [_P, _P, _P, _I, _P, _F, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "noise_gen_vad_bwd": [_P, _P, _I, _L, _P, _P, _P, _P, _I, _I, _P],
    "noise_gen_last_convT_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
})
_lib.lib.noise_gen_pw_wgrad_rows.restype = ctypes.c_int
_lib.lib.noise_gen_pw_wgrad_rows.argtypes = [_L, _I, _I]
_lib.lib.noise_gen_ln_bwd_rows.restype = ctypes.c_int
_lib.lib.noise_gen_ln_bwd_rows.argtypes = [_L]
_lib.lib.noise_gen_last_bwd_rows.restype = ctypes.c_int
_lib.lib.noise_gen_last_bwd_rows.argtypes = []


# ---------------------------------------------------------------------------------------------- thin wrappers
def pw_dgrad(grad, w, npix, Cout, Cin):
    """g_in [npix][Cin] of a 1x1 layer with weight w (Cout, Cin[, 1, 1]) against the lazy gradient `grad`."""
    dz, y, coef3 = grad
    wt = w.reshape(Cout, Cin).t().contiguous()
    g = torch.empty(npix, Cin, device=dz.device, dtype=torch.float32)
    _lib.call("noise_gen_pw_dgrad", ptr(dz), ptr(y), ptr(coef3), ptr(wt), Cout, ptr(g), npix, Cout, Cin, _lib.stream(),
              nbytes=4 * npix * (Cout * (2 if coef3 is not None else 1) + Cin))
    return g


def pw_wgrad(grad, src, npix, Cout, Cin):
    """dW (Cout, Cin) of a 1x1 layer: lazy gradient x lazily normalised input."""
    dz, y, coef3 = grad
    a_y, a_coef, a_relu6 = src
    rows = _lib.lib.noise_gen_pw_wgrad_rows(npix, Cout, Cin)
    slab = torch.empty(rows * Cout * Cin, device=dz.device, dtype=torch.float32)
    dw = torch.empty(Cout, Cin, device=dz.device, dtype=torch.float32)
    _lib.call("noise_gen_pw_wgrad", ptr(dz), ptr(y), ptr(coef3), ptr(a_y), ptr(a_coef), int(bool(a_relu6)), ptr(dw), ptr(slab),
              npix, Cout, Cin, _lib.stream(), nbytes=4 * npix * (Cout * (2 if coef3 is not None else 1) + Cin))
    return dw


def colsum(x, N):
    M = x.numel() // x.shape[-1] if x.dim() > 1 else x.numel()
    ld = x.shape[-1] if x.dim() > 1 else 1
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    slab = torch.empty(256 * N, device=x.device, dtype=torch.float32)
    _lib.call("noise_gen_colsum", ptr(x), ld, M, N, ptr(out), ptr(slab), _lib.stream(), nbytes=4 * M * N)
    return out


def relu_bwd(g, h):
    out = torch.empty_like(g)
    _lib.call("noise_gen_relu_bwd", ptr(g), ptr(h), ptr(out), g.numel(), _lib.stream(), nbytes=12 * g.numel())
    return out


def add_cropped(dst, src):
    """dst (B,Hd,Wd,C)[:, Hd-Hs:, Wd-Ws:] += src (B,Hs,Ws,C), in place."""
    B, Hd, Wd, C = dst.shape
    _, Hs, Ws, _ = src.shape
    _lib.call("noise_gen_add_cropped", ptr(dst), ptr(src), B, Hd, Wd, Hs, Ws, C, _lib.stream(), nbytes=12 * src.numel())


def ln_bwd(g, r, ln, add=None):
    """(d/dr, dgamma, dbeta) of LayerNorm(r) against g (+ add on the input gradient)."""
    M, D = r.shape
    out = torch.empty_like(r)
    dgb = torch.empty(2 * D, device=r.device, dtype=torch.float32)
    rows = _lib.lib.noise_gen_ln_bwd_rows(M)
    slab = torch.empty(rows * 2 * D, device=r.device, dtype=torch.float32)
    _lib.call("noise_gen_ln_bwd", ptr(g), ptr(r), ptr(ln.weight), float(ln.eps), ptr(add), ptr(out), ptr(dgb),
              ctypes.c_void_p(dgb.data_ptr() + 4 * D), ptr(slab), M, D, _lib.stream(), nbytes=4 * M * D * (3 + (add is not None)))
    return out, dgb[:D], dgb[D:]
