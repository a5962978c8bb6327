"""Shape-generic forward of `VADMobileNetV2` (train/models.py:1096-1218 under `Base`, :541-626) for the configurations the
tuned schedule does not cover: repeated and residual inverted-residual blocks (:365, :404-405), 1x3 kernels with stride
(1, 2) on one-row inputs (the waveform-domain train/configs/mobilenetv2.yaml), any number of encoder layers, any model
width / head size, any clip length.  Forward only (eval and train mode); there are no backward kernels for it.

The convolution side runs on the generic kernels written for the once-for-all net (csrc/ofa.hip: lazily normalised NHWC
tensors, per-workgroup BatchNorm partial sums folded by `noise_bn_finalize_fwd`); a 1x3 kernel on a one-row image is the
middle row of a 3x3 kernel whose other rows only ever meet padding.  The transformer bottleneck (:14-225) runs token-major
on `noise_ofa_pw` (the linear layers, fp32 MFMA) and the token kernels of csrc/wave.hip.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, ops
from . import ofa as _ofa  # noqa: F401  (binds the noise_ofa_* entry points)
from ._lib import ptr

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_lib.register({
    "noise_wave_normalize": [_P, _P, _P, _I, _L, _P],
    "noise_wave_final": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "noise_tok_epilogue": [_P, _I, _P, _I, _P, _I, _P, _P, _F, _P, _I, _P, _L, _I, _P],
    "noise_tok_attention": [_P, _P, _P, _I, _P, _F, _P, _I, _I, _I, _I, _I, _F, _P],
    "noise_tok_vad": [_P, _I, _L, _P, _P, _P, _I, _I, _P],
})


def normalize(x):
    """Base._normalize (models.py:614-622): (x / (max|x| + 1e-6), max|x| + 1e-6) per clip."""
    _lib.require_cuda(x)
    B = x.shape[0]
    xin = x.reshape(B, -1).contiguous().float()
    out = torch.empty_like(xin)
    mx = torch.empty(B, device=x.device, dtype=torch.float32)
    _lib.call("noise_wave_normalize", ptr(xin), ptr(out), ptr(mx), B, xin.shape[1], _lib.stream(), nbytes=12 * xin.numel())
    return out.view(x.shape), mx


def finish(y, bias, vad, mx, H, W):
    """`x[:, :, -H:, -W:]` (models.py:610) of the last layer's raw output y (B, Hy, Wy) + its bias, then the gate and the
    de-normalisation of Base.forward (models.py:578-580); a y smaller than (H, W) comes back whole (the padding search
    relies on that)."""
    B, Hy, Wy = y.shape
    Ho, Wo = min(H, Hy), min(W, Wy)
    out = torch.empty(B, 1, Ho, Wo, device=y.device, dtype=torch.float32)
    _lib.call("noise_wave_final", ptr(y), ptr(bias), ptr(vad), ptr(mx), ptr(out), B, Hy, Wy, Ho, Wo, _lib.stream(),
              nbytes=8 * out.numel())
    return out


def _taps3x3(conv, flip):
    """Depthwise / single-channel weight (C, 1, kh, kw) with kh in {1, 3}, kw == 3 as a (C, 3, 3) tensor: a one-row kernel
    is the middle row.  `flip` reverses the taps (a stride-1 transposed convolution as a plain one)."""
    w = conv.weight
    kh, kw = w.shape[-2:]
    if kw != 3 or kh not in (1, 3):
        raise NotImplementedError(f"kernel {kh}x{kw}: the generic forward covers 3x3 and 1x3")
    w3 = torch.zeros(w.shape[0], 3, 3, device=w.device, dtype=torch.float32)
    if kh == 3:
        w3.copy_(w[:, 0])
    else:
        w3[:, 1, :] = w[:, 0, 0, :]
    return w3.flip(-1, -2) if flip else w3


class GenericForward:
    """One forward pass over the module tree of a `VADMobileNetV2`."""

    def __init__(self, model):
        self.m = model
        if model.pad_value != 0:
            raise NotImplementedError("the generic forward pads with zeros (pad_value 0, the waveform models' value)")
        self.max_rows = _lib.lib.noise_ofa_max_rows()
        self.rows = ctypes.c_int(0)

    # ------------------------------------------------------------------ helpers
    def new(self, *shape):
        return torch.empty(*shape, device=self.dev, dtype=torch.float32)

    def _stride(self, conv, H):
        """The layer's stride as one number for the square-kernel entry points."""
        kh = conv.kernel_size[0]
        sh, sw = conv.stride
        if kh == 1:
            if H != 1 or sh != 1:
                raise NotImplementedError(f"a 1x3 kernel on a {H}-row tensor with stride {conv.stride}: the generic forward "
                                          "runs one-row kernels on one-row inputs (the waveform-domain models)")
            return sw
        if sh != sw:
            raise NotImplementedError(f"3x3 kernel with stride {conv.stride}: both strides must be equal")
        return sw

    def bn(self, bn, C, count):
        coef = self.new(5 * C)
        if self.training:
            ops.bn_finalize_fwd(self.slab, self.rows.value, C, count, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                bn.num_batches_tracked, coef, momentum=bn.momentum, eps=bn.eps)
        else:
            ops.bn_coef_eval(C, bn.weight, bn.bias, bn.running_mean, bn.running_var, coef, eps=bn.eps)
        return coef

    def pw(self, src, w, ldw, K, N, npix):
        """out [npix][N] = act(src) [npix][K] x w[:N, :K]^T   (+ the BatchNorm partial sums of the output in the slab)"""
        y, coef, relu6 = src
        if K % 4 or N % 4:
            raise NotImplementedError(f"1x1 layer {K} -> {N}: widths must be multiples of 4")
        out = self.new(npix, N)
        _lib.call("noise_ofa_pw", ptr(y), ptr(coef), int(relu6), ptr(w), ldw, ptr(out), ptr(self.slab), ctypes.byref(self.rows),
                  npix, K, N, _lib.stream(), nbytes=4 * npix * (K + N))
        return out

    def dw(self, src, cna, C, B, Hi, Wi):
        conv, bn = cna[0], cna[1]
        tr = isinstance(conv, nn.ConvTranspose2d)
        s = self._stride(conv, Hi)
        wt = _taps3x3(conv, tr and s == 1).reshape(C, 9).t().contiguous()     # tap-major [9][C]
        Ho = (Hi - 1) * s + 1 if tr else (Hi - 1) // s + 1
        Wo = (Wi - 1) * s + 1 if tr else (Wi - 1) // s + 1
        y, coef, relu6 = src
        out = self.new(B, Ho, Wo, C)
        _lib.call("noise_ofa_dw", ptr(y), ptr(coef), int(relu6), ptr(wt), ptr(out), ptr(self.slab), ctypes.byref(self.rows), B, Hi,
                  Wi, C, 3, s, int(tr), _lib.stream(), nbytes=4 * C * B * (Hi * Wi + Ho * Wo))
        return out, self.bn(bn, C, B * Ho * Wo), Ho, Wo

    def apply(self, y, coef, relu6, res=None, skip=None):
        """act(BatchNorm(y)) [+ res] [+ skip cropped from the end] as a plain NHWC tensor (models.py:404-405, :607-608)."""
        B, Ho, Wo, C = y.shape
        Hs, Ws = (skip.shape[1], skip.shape[2]) if skip is not None else (0, 0)
        if skip is not None and (skip.shape[3] != C or Hs < Ho or Ws < Wo):
            raise RuntimeError(f"decoder tensor {Ho} x {Wo} x {C} against the encoder skip {Hs} x {Ws} x {skip.shape[3]}: the "
                               "sizes do not match (the reference's `x + skip[..., -h:, -w:]` fails the same way)")
        out = self.new(B, Ho, Wo, C)
        _lib.call("noise_ofa_apply", ptr(y), ptr(coef), int(relu6), Ho, Wo, ptr(res), ptr(skip), Hs, Ws, ptr(out), B, Ho, Wo, C,
                  _lib.stream(), nbytes=4 * out.numel() * (2 + (res is not None) + (skip is not None)))
        return out

    def block(self, blk, x, skip=None):
        """InvertedResidual.forward (models.py:348-407) on a plain NHWC tensor; `skip` is what the next decoder block adds
        to this block's output (models.py:607-608), folded into the output pass."""
        B, Hi, Wi, Cin = x.shape
        layers = list(blk.conv)
        src, C = (x, None, False), Cin
        if len(layers) == 4:
            conv, bn = layers[0][0], layers[0][1]
            N = conv.out_channels
            y = self.pw(src, conv.weight, Cin, Cin, N, B * Hi * Wi)
            src, C = (y.view(B, Hi, Wi, N), self.bn(bn, N, B * Hi * Wi), True), N
        y, coef, Ho, Wo = self.dw(src, layers[-3], C, B, Hi, Wi)
        conv, bn = layers[-2], layers[-1]
        N = conv.out_channels
        y = self.pw((y, coef, True), conv.weight, C, C, N, B * Ho * Wo)
        coef = self.bn(bn, N, B * Ho * Wo)
        res = x if blk.use_res_connect else None
        return self.apply(y.view(B, Ho, Wo, N), coef, False, res=res, skip=skip)

    # ------------------------------------------------------------------ transformer bottleneck (models.py:172-225)
    def epilogue(self, y, D, bias=None, relu=False, res=None, ln=None, pre=None):
        M = y.shape[0]
        out = self.new(M, D)
        _lib.call("noise_tok_epilogue", ptr(y), y.shape[1], ptr(bias), int(relu), ptr(res), res.shape[1] if res is not None else 0,
                  ptr(ln.weight) if ln is not None else None, ptr(ln.bias) if ln is not None else None,
                  float(ln.eps) if ln is not None else 0.0, ptr(out), D, ptr(pre), M, D, _lib.stream(),
                  nbytes=4 * M * D * (2 + (res is not None) + (pre is not None)))
        return out

    def linear(self, x, w, K, N):
        return self.pw((x, None, False), w, w.shape[1], K, N, x.shape[0])

    def attention(self, tokens, B, n, D):
        enc = self.m.bottleneck_attention
        if D % 4:
            raise NotImplementedError(f"d_model {D} must be a multiple of 4")
        x = self.epilogue(tokens, D, ln=enc.layer_norm)
        masks = self.m.attn_keep_masks if self.training else None
        p = self.m.attn_dropout if self.training else 0.0
        for li, layer in enumerate(enc.layer_stack):
            sa, ff = layer.slf_attn, layer.pos_ffn
            Hh, dk = sa.n_head, sa.d_k
            HD = Hh * dk
            HDp = (HD + 3) // 4 * 4
            wqkv = torch.zeros(3 * HDp, D, device=self.dev)
            wqkv[0:HD] = sa.w_qs.weight
            wqkv[HDp:HDp + HD] = sa.w_ks.weight
            wqkv[2 * HDp:2 * HDp + HD] = sa.w_vs.weight
            qkv = self.linear(x, wqkv, D, 3 * HDp)
            keep = None
            if masks is not None:
                keep = masks[li].to(device=self.dev, dtype=torch.uint8).contiguous()
                if tuple(keep.shape) != (B, Hh, n, n):
                    raise ValueError(f"attention keep mask {tuple(keep.shape)}, expected {(B, Hh, n, n)}")
            elif p > 0.0:
                keep = torch.empty(B, Hh, n, n, device=self.dev, dtype=torch.uint8).bernoulli_(1.0 - p)   # one byte per weight, no float pass
            o = torch.zeros(B * n, HDp, device=self.dev)
            base = qkv.data_ptr()
            _lib.call("noise_tok_attention", ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * HDp),
                      ctypes.c_void_p(base + 8 * HDp), 3 * HDp, ptr(keep), 1.0 / (1.0 - p) if keep is not None else 1.0, ptr(o),
                      HDp, B, n, Hh, dk, float(dk) ** 0.5, _lib.stream(), nbytes=16 * B * n * HD)
            wfc = torch.zeros(D, HDp, device=self.dev)
            wfc[:, :HD] = sa.fc.weight
            x1 = self.epilogue(self.linear(o, wfc, HDp, D), D, res=x, ln=sa.layer_norm)
            d_in = ff.w_1.out_features
            if d_in % 4:
                raise NotImplementedError(f"d_inner {d_in} must be a multiple of 4")
            h = self.epilogue(self.linear(x1, ff.w_1.weight, D, d_in), d_in, bias=ff.w_1.bias, relu=True)
            x = self.epilogue(self.linear(h, ff.w_2.weight, d_in, D), D, bias=ff.w_2.bias, res=x1, ln=ff.layer_norm)
        vad = self.new(B)
        _lib.call("noise_tok_vad", ptr(x), D, n, ptr(self.m.vad.weight), ptr(self.m.vad.bias), ptr(vad), B, D, _lib.stream())
        return x, vad

    # ------------------------------------------------------------------ Base._forward (models.py:583-612)
    def run(self, xn):
        """Normalised input (B, 1, H, W) -> (raw output of the last layer (B, Hy, Wy) before bias and crop, vad (B,))."""
        m = self.m
        self.dev, self.training = xn.device, m.training
        B, _, H, W = xn.shape
        hp, wp = m.H_padding, m.W_padding
        widths = [mod.num_features for mod in m.modules() if isinstance(mod, nn.BatchNorm2d)]
        enc = m.bottleneck_attention
        widths += [3 * (enc.layer_stack[0].slf_attn.n_head * enc.layer_stack[0].slf_attn.d_k + 3), enc.d_model,
                   enc.layer_stack[0].pos_ffn.w_1.out_features]
        self.slab = self.new(self.max_rows * 2 * max(widths))
        x = xn.reshape(B, H, W).contiguous().float()
        # first layer: 1 -> C, stride 2 (models.py:1151-1159) on the input padded at the top / left (:589)
        conv, bn = m.encoder[0][0], m.encoder[0][1]
        if self._stride(conv, H + hp) != 2:
            raise NotImplementedError("the generic forward's first layer has stride 2")
        C = conv.out_channels
        if C % 4:
            raise NotImplementedError(f"first layer width {C} must be a multiple of 4")
        w9 = _taps3x3(conv, False).contiguous()
        Ho, Wo = (H + hp - 1) // 2 + 1, (W + wp - 1) // 2 + 1
        y = self.new(B, Ho, Wo, C)
        _lib.call("noise_ofa_first_conv", ptr(x), ptr(w9), ptr(y), ptr(self.slab), ctypes.byref(self.rows), B, H, W, hp, wp, C,
                  _lib.stream(), nbytes=4 * (x.numel() + y.numel()))
        x = self.apply(y, self.bn(bn, C, B * Ho * Wo), True)
        downs = [x]
        for blk in list(m.encoder)[1:]:
            x = self.block(blk, x)
            downs.append(x)
        downs = downs[::-1]
        # attention & vad on the tokens of the bottleneck (row-major pixels x channels: NHWC as it lies)
        Bx, h, w, D = x.shape
        tokens, vad = self.attention(x.view(B * h * w, D), B, h * w, D)
        # decoder: every block's input is the running tensor + the mirrored encoder output (models.py:605-609)
        x = self.apply(tokens.view(B, h, w, D), None, False, skip=downs[0])
        blocks = list(m.decoder)
        for i, blk in enumerate(blocks[:-1]):
            x = self.block(blk, x, skip=downs[i + 1])
        conv = blocks[-1][0]
        if self._stride(conv, x.shape[1]) != 2:
            raise NotImplementedError("the generic forward's last layer has stride 2")
        Bx, Hi, Wi, C = x.shape
        w9 = _taps3x3(conv, False).contiguous()
        Hy, Wy = 2 * Hi - 1, 2 * Wi - 1
        y = self.new(B, Hy, Wy)
        _lib.call("noise_ofa_last_convT", ptr(x), None, 0, ptr(w9), ptr(y), ptr(self.slab), ctypes.byref(self.rows), B, Hi, Wi, C,
                  _lib.stream(), nbytes=4 * (x.numel() + y.numel()))
        if self.training:
            m.weights_changed()      # running statistics moved through raw pointers
        return y, vad

    def forward(self, xn):
        """Base._forward: (filters (B, 1, H, W), vad (B, 1)); the first call searches the padding (models.py:585-586)."""
        m = self.m
        B, _, H, W = xn.shape
        if not m.searching and not m.found:
            self.search(xn)
        y, vad = self.run(xn)
        return finish(y, m.decoder[-1][0].bias, None, None, H, W), vad.view(B, 1)

    def search(self, xn):
        """Base._search_padding (models.py:628-639).  The reference grows the padding one step at a time and runs a whole
        `forward` (normalise again, U-Net, gate) per step; the output sizes follow from the strides (`_out_size`), but in
        train mode every one of those forwards also moves the BatchNorm running statistics, so they are run here too."""
        m = self.m
        m.searching = True
        H, W = xn.shape[-2:]

        def probe():
            if m.training:
                self.run(normalize(xn)[0])
        probe()
        while m._out_size(H + m.H_padding, 0) < H:
            m.H_padding += 1
            probe()
        while m._out_size(W + m.W_padding, 1) < W:
            m.W_padding += 1
            probe()
        m.found = True
