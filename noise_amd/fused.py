"""Block-fused bf16 inference of the denoiser U-Net (eval mode, BatchNorm folded into the weights).

Host side of `csrc/fused.hip`: folds the BatchNorm of every conv of train/models.py:348-407 / :299-345 /
:1151-1173 into bf16 weights + fp32 biases in the padded layouts the kernels read, and schedules
`Base._forward` / `BaseSpectral._forward` (train/models.py:583-612, :758-764) as

    head (first conv + block 1)  ->  3 strided blocks  ->  attention (fp32 kernels)  ->  3 transposed blocks
    ->  tail (block 9 + last transposed conv + additive mask)

Activations between kernels are NHWC bf16; skip tensors are read in place with the crop-from-the-end of
train/models.py:608 as an address offset.  There is no CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import ptr

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_lib.register({
    "noise_fused_ir_block": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
})


def _rup(a, b):
    return (a + b - 1) // b * b


def bn_fold(bn):
    """Eval-mode BatchNorm as per-channel (scale, shift)."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    return scale, bn.bias.detach().float() - bn.running_mean.detach().float() * scale


def pad_matrix(w, rows, cols, dtype):
    out = torch.zeros(rows, cols, device=w.device, dtype=torch.float32)
    out[: w.shape[0], : w.shape[1]] = w
    return out.to(dtype).contiguous()


def pad_vector(v, n):
    out = torch.zeros(n, device=v.device, dtype=torch.float32)
    out[: v.numel()] = v
    return out.contiguous()


class IrBlockWeights:
    """Folded, padded parameters of one inverted-residual block with an expansion layer
    (conv = [pw + BN + ReLU6, dw(T) + BN + ReLU6, pw, BN])."""

    def __init__(self, seq, cin, hid, cout):
        s1, sh1 = bn_fold(seq[0][1])
        s2, sh2 = bn_fold(seq[1][1])
        s3, sh3 = bn_fold(seq[3])
        w1 = seq[0][0].weight.detach().float().view(hid, cin) * s1[:, None]
        wd = seq[1][0].weight.detach().float().view(hid, 9) * s2[:, None]
        w3 = seq[2].weight.detach().float().view(cout, hid) * s3[:, None]
        ht, g, ct = _rup(hid, 16), _rup(hid, 8), _rup(cout, 16)
        self.cin, self.hid, self.cout = cin, hid, cout
        self.w1 = pad_matrix(w1, ht, _rup(cin, 32), torch.bfloat16)
        self.b1 = pad_vector(sh1, ht)
        self.wd = pad_matrix(wd.t(), 9, g, torch.float32)          # tap-major
        self.b2 = pad_vector(sh2, g)
        self.w3 = pad_matrix(w3, ct, _rup(hid, 32), torch.bfloat16)
        self.b3 = pad_vector(sh3, ct)


def ir_block(x, skip, crop, wts, out, transposed):
    """x (B,Hi,Wi,Cin) bf16 or fp32 NHWC; skip: same element type, (B,Hs,Ws,Cin), read at offset `crop` = (oy, ox);
    out (B,Ho,Wo,Cout) bf16 or fp32."""
    B, Hi, Wi, Cin = x.shape
    in_f32 = x.dtype == torch.float32
    if skip is not None:
        assert skip.dtype == x.dtype and skip.shape[0] == B and skip.shape[3] == Cin
    Hs, Ws = (skip.shape[1], skip.shape[2]) if skip is not None else (0, 0)
    oy, ox = crop if skip is not None else (0, 0)
    _lib.call("noise_fused_ir_block", ptr(x), ptr(skip), int(in_f32), Hs, Ws, oy, ox, ptr(wts.w1), ptr(wts.b1),
              ptr(wts.wd), ptr(wts.b2), ptr(wts.w3), ptr(wts.b3), ptr(out), int(out.dtype == torch.float32),
              int(transposed), B, Hi, Wi, Cin, wts.hid, wts.cout, _lib.stream(),
              nbytes=x.numel() * x.element_size() * (2 if skip is not None else 1) + out.numel() * out.element_size())
