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
import os

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ptr
from .generic import GenericForward, _taps3x3, finish, normalize

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_lib.register({
    "noise_gen_slab_reduce": [_P, _I, _L, _P, _P],
    "noise_gen_pw_dgrad": [_P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _L, _I, _I, _P],
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
def pw_dgrad(grad, w, npix, Cout, Cin, epi=None, slab=None):
    """g_in [npix][Cin] of a 1x1 layer with weight w (Cout, Cin[, 1, 1]) against the lazy gradient `grad`.  With
    `epi = (y, coef, relu6)` of the BatchNorm (+ReLU6) that produced the layer's input, g_in comes back masked and
    (g_in, rows) is returned: `slab` then holds `rows` rows of that BatchNorm's backward sums (2 Cin floats each)."""
    dz, y, coef3 = grad
    wt = w.reshape(Cout, Cin).t().contiguous()
    g = torch.empty(npix, Cin, device=dz.device, dtype=torch.float32)
    rows = ctypes.c_int(0)
    ey, ecoef, erelu = epi if epi is not None else (None, None, False)
    _lib.call("noise_gen_pw_dgrad", ptr(dz), ptr(y), ptr(coef3), ptr(wt), Cout, ptr(g), ptr(ey), ptr(ecoef), int(bool(erelu)),
              ptr(slab), ctypes.byref(rows), npix, Cout, Cin, _lib.stream(),
              nbytes=4 * npix * (Cout * (2 if coef3 is not None else 1) + Cin * (2 if epi is not None else 1)))
    return (g, rows.value) if epi is not None else g


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


class GenericTrain(GenericForward):
    """Forward with a tape + backward for one batch; `grads` maps every parameter to its gradient."""

    def __init__(self, model):
        super().__init__(model)
        self.tape = []
        self.grads = {}
        self.g = {}          # id(tensor) -> [tensor, gradient]: gradients of materialised tensors
        self.scratch = None

    # ------------------------------------------------------------------ gradient bookkeeping
    def acc(self, t, g, own=False):
        """Add g (the shape of t, or smaller: cropped from the end, models.py:608) to the gradient of tensor t.  `own`: g
        belongs to nobody else and may become the accumulator."""
        if g.dim() == 2:
            g = g.view(t.shape)
        cur = self.g.get(id(t))
        if cur is None:
            if g.shape == t.shape:
                self.g[id(t)] = [t, g if own else g.clone()]
            else:
                z = torch.zeros_like(t)
                add_cropped(z, g)
                self.g[id(t)] = [t, z]
        else:
            acc4 = cur[1] if cur[1].dim() == 4 else cur[1].view(1, 1, *cur[1].shape)
            add_cropped(acc4, g if g.dim() == 4 else g.view(1, 1, *g.shape))

    def take(self, t):
        return self.g.pop(id(t))[1]

    def bn_finalize(self, dz, slab, rows, row_stride, y, coef, bn, npix, C):
        """The rest of a BatchNorm backward whose masked gradient `dz` and two sums a consumer's epilogue already left."""
        dgb = self.new(2 * C)
        coef3 = self.new(6 * C)
        ops.bn_bwd_finalize(slab, rows, row_stride, coef, bn.weight, dgb[:C], dgb[C:], coef3, npix, C)
        self.grads[bn.weight], self.grads[bn.bias] = dgb[:C], dgb[C:]
        return dz, y, coef3

    def bn_bwd(self, g_act, y, coef, relu6, bn, npix, C):
        """BatchNorm (+ReLU6) backward: gradient of the activated output -> lazy gradient of the raw convolution output."""
        dz = self.new(npix, C)
        dgb = self.new(2 * C)
        coef3 = self.new(6 * C)
        ops.bn_bwd(g_act, None, y, coef, relu6, bn.weight, dz, dgb[:C], dgb[C:], coef3, self.scratch, npix, C)
        self.grads[bn.weight], self.grads[bn.bias] = dgb[:C], dgb[C:]
        return dz, y, coef3

    @staticmethod
    def _dw_grad(conv, dw9):
        """(C,3,3) tap gradients -> the shape of the layer's weight (a 1x3 kernel is the middle row)."""
        return dw9.view_as(conv.weight) if conv.weight.shape[-2] == 3 else dw9[:, 1, :].reshape(conv.weight.shape)

    # ------------------------------------------------------------------ layers
    def block(self, blk, x, skip=None):
        B, Hi, Wi, Cin = x.shape
        layers = list(blk.conv)
        src, C, exp = (x, None, False), Cin, None
        npi = B * Hi * Wi
        if len(layers) == 4:
            conv1, bn1 = layers[0][0], layers[0][1]
            N1 = conv1.out_channels
            y1 = self.pw(src, conv1.weight, Cin, Cin, N1, npi)
            coef1 = self.bn(bn1, N1, npi)
            exp = (conv1, bn1, y1, coef1)
            src, C = (y1.view(B, Hi, Wi, N1), coef1, True), N1
        dwl = layers[-3]
        y2, coef2, Ho, Wo = self.dw(src, dwl, C, B, Hi, Wi)
        npo = B * Ho * Wo
        conv3, bn3 = layers[-2], layers[-1]
        N = conv3.out_channels
        y3 = self.pw((y2, coef2, True), conv3.weight, C, C, N, npo)
        coef3 = self.bn(bn3, N, npo)
        res = x if blk.use_res_connect else None
        out = self.apply(y3.view(B, Ho, Wo, N), coef3, False, res=res, skip=skip)

        def backward():
            G = self.take(out)
            if skip is not None:
                self.acc(skip, G)
            if res is not None:
                self.acc(x, G)
            g3 = self.bn_bwd(G, y3, coef3, False, bn3, npo, N)
            self.grads[conv3.weight] = pw_wgrad(g3, (y2, coef2, True), npo, N, C).view_as(conv3.weight)
            # every backward kernel finishes the BatchNorm (+ReLU6) backward of the layer that produced its input: it
            # masks its output and leaves the two per-channel sums, so those BatchNorms cost no pass of their own
            dz2, rows = pw_dgrad(g3, conv3.weight, npo, N, C, epi=(y2, coef2, True), slab=self.slab)
            g2 = self.bn_finalize(dz2, self.slab, rows, 2 * C, y2, coef2, dwl[1], npo, C)
            conv2 = dwl[0]
            tr = isinstance(conv2, nn.ConvTranspose2d)
            s = self._stride(conv2, Hi)
            w9 = _taps3x3(conv2, False).contiguous()
            ga1, dw9 = self.new(B, Hi, Wi, C), self.new(C, 3, 3)
            # One-row tensors go to the one-row kernels of noise_dw3x3_bwd (several consecutive pixels per thread, the
            # layer's raw output rebuilt from the input): stride 1 as it is, stride 2 through unequal strides (the row
            # stride is immaterial with one row).  NOISE_GEN_ROWS=1 (diagnostic) hands a stride-1 layer to the 2-D column
            # walkers as ONE image of B rows instead -- a 1 x 3 kernel couples no rows and the other two rows of the
            # embedded 3 x 3 kernel are zero -- which measures slower (23.7 against 21.7 ms per step of the waveform
            # network).
            rows_as_image = Hi == 1 and s == 1 and conv2.kernel_size[0] == 1 and os.environ.get("NOISE_GEN_ROWS") == "1"
            Bk, Hk = (1, B) if rows_as_image else (B, Hi)
            epi = (exp[3], True) if exp is not None else None
            strides = (1, s) if (Hi == 1 and s == 2) else (s, s)
            rows = ops.dw3x3_bwd(g2, src, None, w9, ga1, dw9, self.scratch, tr, Bk, Hk, Wi, C, strides, None, epi)
            self.grads[conv2.weight] = self._dw_grad(conv2, dw9)
            if exp is not None:
                conv1, bn1, y1, coef1 = exp
                g1 = self.bn_finalize(ga1, self.scratch[9 * C:], rows, 11 * C, y1, coef1, bn1, npi, C)
                self.grads[conv1.weight] = pw_wgrad(g1, (x, None, False), npi, C, Cin).view_as(conv1.weight)
                self.acc(x, pw_dgrad(g1, conv1.weight, npi, C, Cin), own=True)
            else:
                self.acc(x, ga1, own=True)
        self.tape.append(backward)
        return out

    def attention(self, tokens, B, n, D):
        m = self.m
        enc = m.bottleneck_attention
        if D % 4:
            raise NotImplementedError(f"d_model {D} must be a multiple of 4")
        M = B * n
        x = self.epilogue(tokens, D, ln=enc.layer_norm)
        masks = m.attn_keep_masks
        p = m.attn_dropout
        saved = []
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
            ks = 1.0 / (1.0 - p) if keep is not None else 1.0
            o = torch.zeros(M, HDp, device=self.dev)
            base = qkv.data_ptr()
            _lib.call("noise_tok_attention", ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * HDp),
                      ctypes.c_void_p(base + 8 * HDp), 3 * HDp, ptr(keep), ks, ptr(o), HDp, B, n, Hh, dk, float(dk) ** 0.5,
                      _lib.stream(), nbytes=16 * M * HD)
            wfc = torch.zeros(D, HDp, device=self.dev)
            wfc[:, :HD] = sa.fc.weight
            r1 = self.new(M, D)
            x1 = self.epilogue(self.linear(o, wfc, HDp, D), D, res=x, ln=sa.layer_norm, pre=r1)
            d_in = ff.w_1.out_features
            if d_in % 4:
                raise NotImplementedError(f"d_inner {d_in} must be a multiple of 4")
            h = self.epilogue(self.linear(x1, ff.w_1.weight, D, d_in), d_in, bias=ff.w_1.bias, relu=True)
            r2 = self.new(M, D)
            x2 = self.epilogue(self.linear(h, ff.w_2.weight, d_in, D), D, bias=ff.w_2.bias, res=x1, ln=ff.layer_norm, pre=r2)
            saved.append((layer, x, wqkv, qkv, keep, ks, o, wfc, r1, x1, h, r2, HD, HDp, d_in))
            x = x2
        vad = self.new(B)
        _lib.call("noise_tok_vad", ptr(x), D, n, ptr(m.vad.weight), ptr(m.vad.bias), ptr(vad), B, D, _lib.stream())
        xf = x

        def backward():
            gx = self.take(xf)                       # gradient of the token matrix from the decoder (a private copy)
            dw, db = self.new(D), self.new(1)
            _lib.call("noise_gen_vad_bwd", ptr(xf), ptr(gx), D, n, ptr(m.vad.weight), ptr(self.dvad), ptr(dw), ptr(db), B, D,
                      _lib.stream())
            self.grads[m.vad.weight], self.grads[m.vad.bias] = dw.view_as(m.vad.weight), db
            for layer, xin, wqkv, qkv, keep, ks, o, wfc, r1, x1, h, r2, HD, HDp, d_in in reversed(saved):
                sa, ff = layer.slf_attn, layer.pos_ffn
                Hh, dk = sa.n_head, sa.d_k
                g_r2, self.grads[ff.layer_norm.weight], self.grads[ff.layer_norm.bias] = ln_bwd(gx, r2, ff.layer_norm)
                plain = (g_r2, None, None)
                self.grads[ff.w_2.bias] = colsum(g_r2, D)
                self.grads[ff.w_2.weight] = pw_wgrad(plain, (h, None, False), M, D, d_in)
                g_hpre = relu_bwd(pw_dgrad(plain, ff.w_2.weight, M, D, d_in), h)
                plain1 = (g_hpre, None, None)
                self.grads[ff.w_1.bias] = colsum(g_hpre, d_in)
                self.grads[ff.w_1.weight] = pw_wgrad(plain1, (x1, None, False), M, d_in, D)
                g_x1 = pw_dgrad(plain1, ff.w_1.weight, M, d_in, D)
                add_cropped(g_x1.view(1, 1, M, D), g_r2.view(1, 1, M, D))           # + the residual branch
                g_r1, self.grads[sa.layer_norm.weight], self.grads[sa.layer_norm.bias] = ln_bwd(g_x1, r1, sa.layer_norm)
                plain2 = (g_r1, None, None)
                self.grads[sa.fc.weight] = pw_wgrad(plain2, (o, None, False), M, D, HDp)[:, :HD].contiguous()
                g_o = pw_dgrad(plain2, wfc, M, D, HDp)
                dqkv = torch.zeros_like(qkv)
                stats = self.new(B * Hh * n * 3)
                base, dbase = qkv.data_ptr(), dqkv.data_ptr()
                _lib.call("noise_gen_attn_bwd", ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * HDp),
                          ctypes.c_void_p(base + 8 * HDp), 3 * HDp, ptr(keep), ks, ptr(o), ptr(g_o), HDp, ctypes.c_void_p(dbase),
                          ctypes.c_void_p(dbase + 4 * HDp), ctypes.c_void_p(dbase + 8 * HDp), ptr(stats), B, n, Hh, dk,
                          float(dk) ** 0.5, _lib.stream(), nbytes=32 * M * HD)
                plain3 = (dqkv, None, None)
                dwqkv = pw_wgrad(plain3, (xin, None, False), M, 3 * HDp, D)
                self.grads[sa.w_qs.weight] = dwqkv[0:HD].contiguous()
                self.grads[sa.w_ks.weight] = dwqkv[HDp:HDp + HD].contiguous()
                self.grads[sa.w_vs.weight] = dwqkv[2 * HDp:2 * HDp + HD].contiguous()
                gx = pw_dgrad(plain3, wqkv, M, 3 * HDp, D)
                add_cropped(gx.view(1, 1, M, D), g_r1.view(1, 1, M, D))
            g_tok, self.grads[enc.layer_norm.weight], self.grads[enc.layer_norm.bias] = ln_bwd(gx, tokens, enc.layer_norm)
            self.acc(self.bottleneck, g_tok, own=True)
        self.tape.append(backward)
        return x, vad

    # ------------------------------------------------------------------ whole network
    def run(self, xn):
        """Train-mode forward with the tape filled: (raw last-layer output (B, Hy, Wy), vad (B,))."""
        m = self.m
        self.dev, self.training = xn.device, True
        B, _, H, W = xn.shape
        hp, wp = m.H_padding, m.W_padding
        widths = [mod.num_features for mod in m.modules() if isinstance(mod, nn.BatchNorm2d)]
        enc = m.bottleneck_attention
        widths += [3 * (enc.layer_stack[0].slf_attn.n_head * enc.layer_stack[0].slf_attn.d_k + 3), enc.d_model,
                   enc.layer_stack[0].pos_ffn.w_1.out_features]
        self.slab = self.new(self.max_rows * 2 * max(widths))
        self.scratch = self.new(_lib.lib.noise_conv_max_blocks() * 12 * max(widths))
        x0 = xn.reshape(B, H, W).contiguous().float()
        conv, bn0 = m.encoder[0][0], m.encoder[0][1]
        if self._stride(conv, H + hp) != 2:
            raise NotImplementedError("the generic forward's first layer has stride 2")
        C0 = conv.out_channels
        w9 = _taps3x3(conv, False).contiguous()
        Ho, Wo = (H + hp - 1) // 2 + 1, (W + wp - 1) // 2 + 1
        y0 = self.new(B, Ho, Wo, C0)
        _lib.call("noise_ofa_first_conv", ptr(x0), ptr(w9), ptr(y0), ptr(self.slab), ctypes.byref(self.rows), B, H, W, hp, wp, C0,
                  _lib.stream(), nbytes=4 * (x0.numel() + y0.numel()))
        coef0 = self.bn(bn0, C0, B * Ho * Wo)
        a0 = self.apply(y0, coef0, True)
        first_conv = conv

        def first_backward():
            G = self.take(a0)
            g0 = self.bn_bwd(G, y0, coef0, True, bn0, B * Ho * Wo, C0)
            dw9 = self.new(C0, 3, 3)
            one = torch.full((B,), 1.0 - 1e-6, device=self.dev)       # the kernel divides by (absmax + 1e-6)
            ops.first_conv_wgrad(g0, x0, one, dw9, self.scratch, B, H, W, hp, wp, 0.0, C0, (2, 2))
            self.grads[first_conv.weight] = self._dw_grad(first_conv, dw9)
        self.tape.append(first_backward)
        x = a0
        downs = [x]
        for blk in list(m.encoder)[1:]:
            x = self.block(blk, x)
            downs.append(x)
        downs = downs[::-1]
        Bx, h, w, D = x.shape
        self.bottleneck = x
        tokens, vad = self.attention(x.view(B * h * w, D), B, h * w, D)
        tok4 = tokens.view(B, h, w, D)
        xd = self.apply(tok4, None, False, skip=downs[0])

        def join_backward():
            G = self.take(xd)
            self.acc(downs[0], G)
            self.g[id(tokens)] = [tokens, G.view(B * h * w, D)]     # consumed (and modified) by the attention backward
        self.tape.append(join_backward)
        x = xd
        blocks = list(m.decoder)
        for i, blk in enumerate(blocks[:-1]):
            x = self.block(blk, x, skip=downs[i + 1])
        last = blocks[-1][0]
        if self._stride(last, x.shape[1]) != 2:
            raise NotImplementedError("the generic forward's last layer has stride 2")
        Bx, Hi, Wi, C = x.shape
        wl = _taps3x3(last, False).contiguous()
        Hy, Wy = 2 * Hi - 1, 2 * Wi - 1
        y = self.new(B, Hy, Wy)
        _lib.call("noise_ofa_last_convT", ptr(x), None, 0, ptr(wl), ptr(y), ptr(self.slab), ctypes.byref(self.rows), B, Hi, Wi, C,
                  _lib.stream(), nbytes=4 * (x.numel() + y.numel()))
        xl = x

        def last_backward():
            g = self.new(B, Hi, Wi, C)
            dw9 = self.new(C, 3, 3)
            slab = self.new(_lib.lib.noise_gen_last_bwd_rows() * 9 * C)
            _lib.call("noise_gen_last_convT_bwd", ptr(self.dout), ptr(xl), ptr(wl), ptr(g), ptr(dw9), ptr(slab), B, Hi, Wi, C,
                      _lib.stream(), nbytes=4 * (2 * xl.numel() + self.dout.numel()))
            self.grads[last.weight] = self._dw_grad(last, dw9)
            self.grads[last.bias] = colsum(self.dout.view(-1, 1), 1)
            self.acc(xl, g, own=True)
        self.tape.append(last_backward)
        m.weights_changed()
        return y, vad

    def backward(self, dout, dvad):
        """dout: gradient of the last layer's raw output (B, Hy, Wy); dvad: gradient of the VAD logits (B,)."""
        self.dout, self.dvad = dout.contiguous(), dvad.contiguous().float()
        for fn in reversed(self.tape):
            fn()
        self.tape.clear()
        assert not self.g, "gradients left without a consumer"
        return self.grads


def shared_step(model, noisy, clean):
    """Base._shared_step (train/models.py:641-679) with gradients: (losses [total, waveform, spectral, vad], {param: grad})."""
    from .losses import mrstft_loss_and_grad, smooth_l1_and_grad
    hp = model.hparams
    S = model.SEGMENT
    noisy = noisy.reshape(-1, 1, 1, S)
    clean = clean.reshape(-1, 1, 1, S).contiguous().float()
    B = noisy.shape[0]
    xn, mx = normalize(noisy)
    run = GenericTrain(model)
    if not model.searching and not model.found:
        run.search(xn)
        run.tape.clear()
    y, vad = run.run(xn)
    Hy, Wy = y.shape[1:]
    enhanced = finish(y, model.decoder[-1][0].bias, None, mx, 1, S)                       # filters * max (no gate)
    l1, g_l1 = smooth_l1_and_grad(enhanced, clean, 0.5, want_grad=True)
    mr, _, g_mr = mrstft_loss_and_grad(enhanced, clean, tuple(hp.fft_sizes), tuple(hp.hop_lengths), tuple(hp.win_lengths),
                                       want_grad=True)
    label = torch.empty(B, device=clean.device)
    _lib.call("noise_vad_label", ptr(clean), ptr(label), B, S, 160, _lib.stream())
    z = vad.double()
    vad_loss = torch.nn.functional.binary_cross_entropy_with_logits(z, label.double())
    dvad = ((torch.sigmoid(z) - label.double()) / B).float()
    lam = float(hp.mr_stft_lambda)
    d_enh = (g_l1.reshape(B, S).double() + lam * g_mr.reshape(B, S).double()).float()
    d_filt = finish(d_enh.view(B, 1, S).contiguous(), None, None, mx, 1, S)                  # chain through `* max`
    dout = torch.zeros(B, Hy, Wy, device=y.device)
    dout[:, Hy - 1:, Wy - S:] = d_filt.view(B, 1, S)
    grads = run.backward(dout, dvad)
    losses = torch.stack([l1.double() + lam * mr.double() + vad_loss, l1.double(), lam * mr.double(), vad_loss]).float()
    return losses, grads


class StepFn(torch.autograd.Function):
    """loss = shared_step(noisy, clean): the backward pass runs inside forward (a training step always back-propagates) and
    its gradients are handed to autograd in backward, so `loss.backward()` fills every parameter's `.grad`."""

    @staticmethod
    def forward(ctx, model, noisy, clean, *params):
        losses, grads = shared_step(model, noisy, clean)
        missing = [n for n, p in model.named_parameters() if p not in grads]
        assert not missing, f"no gradient for {missing}"
        ctx.grads = [grads[p].reshape(p.shape).contiguous() for p in params]
        if getattr(model, "grad_sync", None) is not None:     # data parallel: mean over ranks (dist.make_grad_mean)
            model.grad_sync(ctx.grads)
        all4 = losses.clone()
        ctx.mark_non_differentiable(all4)
        return losses[0].clone(), all4

    @staticmethod
    def backward(ctx, g_loss, g_all):
        return (None, None, None) + tuple(g * g_loss for g in ctx.grads)
