"""Block-fused bf16 inference of the denoiser U-Net (eval mode, BatchNorm folded into the weights).

Host side of `csrc/fused.hip`: folds the BatchNorm of every conv of train/models.py:348-407 / :299-345 /
:1151-1173 into half-precision weights + biases in the padded layouts the kernels read, and schedules
`Base._forward` / `BaseSpectral._forward` (train/models.py:583-612, :758-764) as

    head (first conv + block 1)  ->  3 strided blocks  ->  attention (fp32 kernels)  ->  3 transposed blocks
    ->  tail (block 9 + last transposed conv + additive mask)

Activations between kernels are NHWC bf16; skip tensors are read in place with the crop-from-the-end of
train/models.py:608 as an address offset.  There is no CPU path.
"""
import ctypes

import torch

from . import _lib
from . import engine  # noqa: F401  (binds the attention entry points used below)
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
        # the kernels keep hidden activations divided by 6 (ReLU6 = clamp to [0, 1]): 1/6 goes into what produces them
        # (w1, b1, b2; the depthwise weights map h/6 to h/6 unchanged), 6 into what consumes them (w3)
        w1 = seq[0][0].weight.detach().float().view(hid, cin) * s1[:, None] / 6.0
        sh1, sh2 = sh1 / 6.0, sh2 / 6.0
        wd = seq[1][0].weight.detach().float().view(hid, 9) * s2[:, None]
        w3 = seq[2].weight.detach().float().view(cout, hid) * s3[:, None] * 6.0
        ht, g, ct = _rup(hid, 16), _rup(hid, 8), _rup(cout, 16)
        self.cin, self.hid, self.cout = cin, hid, cout
        self.w1 = pad_matrix(w1, ht, _rup(cin, 32), torch.float16)
        self.b1 = pad_vector(sh1, ht)
        self.wd = pad_matrix(wd.t(), 9, g, torch.float16)          # tap-major
        self.b2 = pad_vector(sh2, g).half()
        self.w3 = pad_matrix(w3, ct, _rup(hid, 32), torch.float16)
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


_lib.register({
    "noise_attn_core_fwd_f16": [_P, _P, ctypes.c_long, _I, _I, _I, _I, _P],
    "noise_fused_head": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "noise_fused_tail": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P,
                         _I, _I, _I, _I, _I, _P],
})


class HeadWeights:
    """First conv (1->32, BN, ReLU6) + block 1 (depthwise 32 + BN + ReLU6, 1x1 32->16 + BN)."""

    def __init__(self, first, seq):
        s0, sh0 = bn_fold(first[1])
        w0 = first[0].weight.detach().float().view(32, 9) * s0[:, None] / 6.0     # hidden activations are kept / 6
        sh0 = sh0 / 6.0
        # reduction slot k = 8q + 2t + s is tap 4q + t (zero weight past tap 8), s = 0 for the input rounded to half and
        # s = 1 for its remainder: lane group q of the MFMA B operand then reads only its own four taps
        wk = torch.zeros(32, 32, dtype=torch.float32, device=w0.device)
        for q in range(4):
            for t in range(4):
                if 4 * q + t < 9:
                    wk[:, 8 * q + 2 * t] = w0[:, 4 * q + t]
                    wk[:, 8 * q + 2 * t + 1] = w0[:, 4 * q + t]
        self.w0 = pad_matrix(wk, 32, 32, torch.float16)
        self.b0 = sh0.contiguous()
        s2, sh2 = bn_fold(seq[0][1])
        s3, sh3 = bn_fold(seq[2])
        self.wd = (seq[0][0].weight.detach().float().view(32, 9) * s2[:, None]).t().contiguous().half()
        self.b2 = (sh2 / 6.0).contiguous().half()
        self.w3 = pad_matrix(seq[1].weight.detach().float().view(16, 32) * s3[:, None] * 6.0, 16, 32, torch.float16)
        self.b3 = sh3.contiguous()


class TailWeights:
    """Block 9 (depthwise ConvTranspose2d 16 s1 + BN + ReLU6, 1x1 16->32 + BN) + last ConvTranspose2d 32->1 with bias."""

    def __init__(self, seq, last):
        s2, sh2 = bn_fold(seq[0][1])
        s3, sh3 = bn_fold(seq[2])
        # the depthwise input (a sum of two block outputs) is not scaled, its output is kept / 6
        self.wd = (seq[0][0].weight.detach().float().view(16, 9) * s2[:, None] / 6.0).t().contiguous().half()
        self.b2 = (sh2 / 6.0).contiguous().half()
        self.w3 = pad_matrix(seq[1].weight.detach().float().view(32, 16) * s3[:, None] * 6.0, 32, 32, torch.float16)
        self.b3 = sh3.contiguous()
        self.wl = pad_matrix(last.weight.detach().float().view(32, 9).t(), 16, 32, torch.float16)   # [tap][channel]
        self.bias = float(last.bias.detach().float().item())


SUPPORTED_PLAN = ([dict(kind="first", cin=1, cout=32, stride=(2, 2)),
                   dict(kind="ir", cin=32, cout=16, t=1, stride=(1, 1), hidden=32),
                   dict(kind="ir", cin=16, cout=24, t=3, stride=(2, 2), hidden=48),
                   dict(kind="ir", cin=24, cout=32, t=3, stride=(2, 2), hidden=72),
                   dict(kind="ir", cin=32, cout=64, t=3, stride=(2, 2), hidden=96)])


class FusedDenoiser:
    """bf16 block-fused eval forward of one VADMobileNetV2 for one (batch, H, W) geometry."""

    def __init__(self, net, B, H, W, device):
        enc, dec = net.block_plan()
        if enc != SUPPORTED_PLAN or net.n_head * net.d_k > 64:
            raise NotImplementedError("the block-fused bf16 path serves the layer_config of train/configs/spectralv2.yaml; "
                                      "other configurations run through the per-layer fp32 kernels (inference_dtype='fp32')")
        self.net, self.B, self.H, self.W, self.device = net, B, H, W, device
        self.hp, self.wp = net.H_padding, net.W_padding
        bf, f32 = dict(device=device, dtype=torch.bfloat16), dict(device=device, dtype=torch.float32)
        down = lambda n: (n - 1) // 2 + 1
        up = lambda n: 2 * n - 1
        H1, W1 = down(H + self.hp), down(W + self.wp)
        sizes = [(H1, W1)]
        for _ in range(3):
            sizes.append((down(sizes[-1][0]), down(sizes[-1][1])))
        self.a0 = torch.empty(B, H1, W1, 32, **bf)
        self.a1 = torch.empty(B, H1, W1, 16, **bf)
        self.a2 = torch.empty(B, *sizes[1], 24, **bf)
        self.a3 = torch.empty(B, *sizes[2], 32, **bf)
        self.a4 = torch.empty(B, *sizes[3], 64, **f32)
        h, w = sizes[3]
        self.n_tok, self.ntok = h * w, B * h * w
        # decoder geometry and the crop-from-the-end offsets of the skip tensors (train/models.py:608)
        self.crops = []
        skips = [self.a4, self.a3, self.a2, self.a1, self.a0]
        cur = (h, w)
        for i, sk in enumerate(skips):
            oy, ox = sk.shape[1] - cur[0], sk.shape[2] - cur[1]
            if oy < 0 or ox < 0:
                raise RuntimeError(f"decoder block {i}: skip {sk.shape[1]}x{sk.shape[2]} is smaller than the decoder tensor "
                                   f"{cur[0]}x{cur[1]} (the reference's `x + skip[..., -h:, -w:]` fails to broadcast as well)")
            self.crops.append((oy, ox))
            if i < 3:
                cur = (up(cur[0]), up(cur[1]))
        self.d0 = torch.empty(B, up(h), up(w), 32, **bf)
        self.d1 = torch.empty(B, up(up(h)), up(up(w)), 24, **bf)
        self.d2 = torch.empty(B, *cur, 16, **bf)
        if up(cur[0]) < H or up(cur[1]) < W:
            raise RuntimeError(f"U-Net output {up(cur[0])}x{up(cur[1])} is smaller than its input {H}x{W}: "
                               "H_padding/W_padding do not fit this input size")
        # attention buffers (fp32 kernels of csrc/attn.hip)
        D, hd, nt = 64, net.n_head * net.d_k, self.ntok
        self.xhat0 = torch.empty(nt, D, **f32)
        self.rstd0 = torch.empty(nt, **f32)
        self.qkv = torch.empty(nt, 3 * hd, **f32)
        self.att_o = torch.empty(nt, hd, **f32)
        self.xhat1 = torch.empty(nt, D, **f32)
        self.rstd1 = torch.empty(nt, **f32)
        self.hdn = torch.empty(nt, D, **f32)
        self.xhat2 = torch.empty(nt, D, **f32)
        self.rstd2 = torch.empty(nt, **f32)
        self.x3 = torch.empty(nt, D, **f32)
        self.vad = torch.empty(B, **f32)
        self.enh = torch.empty(B, H, W, **f32)
        self.pinned = 0
        self._wkey = None
        self._wtensors = None

    def weights_key(self):
        """Identity of the values the folded weight images were made from: autograd versions catch in-place tensor
        writes, the net's update counter catches writes through raw device pointers (fused AdamW, running statistics
        of a training forward, state_dict loads), the data pointer catches reallocation."""
        net = self.net
        first = next(net.parameters())
        ts = self._wtensors
        if ts is None or ts[0] is not first:       # tensor objects replaced (.to(), a new Parameter): list them again
            ts = self._wtensors = list(net.parameters()) + list(net.buffers())
        return (net._weights_epoch, first.data_ptr()) + tuple(t._version for t in ts)

    def _weights(self):
        net = self.net
        key = self.weights_key()
        if self._wkey != key:
            e, d = net.encoder, net.decoder
            self.head = HeadWeights(e[0], e[1].conv)
            self.enc = [IrBlockWeights(e[i].conv, c, hdn, o) for i, (c, hdn, o) in ((2, (16, 48, 24)), (3, (24, 72, 32)), (4, (32, 96, 64)))]
            self.dec = [IrBlockWeights(d[i].conv, c, hdn, o) for i, (c, hdn, o) in ((0, (64, 192, 32)), (1, (32, 96, 24)), (2, (24, 72, 16)))]
            self.tail = TailWeights(d[3].conv, d[4][0])
            self._wkey = key

    def forward(self, logmag, absmax, want_filt=False):
        """logmag (B,H,W) fp32 raw log-magnitudes, absmax (B,) -> enhanced log-magnitudes (B,H,W) fp32, vad (B,)."""
        self._weights()
        net, B, H, W = self.net, self.B, self.H, self.W
        st = _lib.stream()
        hw = self.head
        _lib.call("noise_fused_head", ptr(logmag), ptr(absmax), ptr(hw.w0), ptr(hw.b0), ptr(hw.wd), ptr(hw.b2), ptr(hw.w3),
                  ptr(hw.b3), ptr(self.a0), ptr(self.a1), B, H, W, self.hp, self.wp, float(net.pad_value), st,
                  nbytes=4 * B * H * W + 2 * (self.a0.numel() + self.a1.numel()))
        ir_block(self.a1, None, None, self.enc[0], self.a2, False)
        ir_block(self.a2, None, None, self.enc[1], self.a3, False)
        ir_block(self.a3, None, None, self.enc[2], self.a4, False)
        att = net.bottleneck_attention
        ly = att.layer_stack[0]
        sa, ff = ly.slf_attn, ly.pos_ffn
        hd = net.n_head * net.d_k
        dims = (self.ntok, self.n_tok, 64, net.n_head, net.d_k)
        _lib.call("noise_attn_pre_fwd", ptr(self.a4), None, 0, ptr(att.layer_norm.weight), ptr(att.layer_norm.bias),
                  ptr(sa.w_qs.weight), ptr(sa.w_ks.weight), ptr(sa.w_vs.weight), ptr(self.xhat0), ptr(self.rstd0),
                  ptr(self.qkv), *dims, st, nbytes=4 * self.ntok * (2 * 64 + 3 * hd))
        _lib.call("noise_attn_core_fwd_f16", ptr(self.qkv), ptr(self.att_o), *dims, st, nbytes=4 * self.ntok * 4 * hd)
        _lib.call("noise_attn_post_fwd", ptr(self.att_o), ptr(self.xhat0), ptr(att.layer_norm.weight),
                  ptr(att.layer_norm.bias), ptr(sa.fc.weight), ptr(sa.layer_norm.weight), ptr(sa.layer_norm.bias),
                  ptr(ff.w_1.weight), ptr(ff.w_1.bias), ptr(ff.w_2.weight), ptr(ff.w_2.bias), ptr(ff.layer_norm.weight),
                  ptr(ff.layer_norm.bias), ptr(net.vad.weight), ptr(net.vad.bias), ptr(self.xhat1), ptr(self.rstd1),
                  ptr(self.hdn), ptr(self.xhat2), ptr(self.rstd2), ptr(self.x3), ptr(self.vad), *dims, st,
                  nbytes=4 * self.ntok * (hd + 5 * 64))
        x3 = self.x3.view(self.a4.shape)
        ir_block(x3, self.a4, self.crops[0], self.dec[0], self.d0, True)
        ir_block(self.d0, self.a3, self.crops[1], self.dec[1], self.d1, True)
        ir_block(self.d1, self.a2, self.crops[2], self.dec[2], self.d2, True)
        tw = self.tail
        filt = torch.empty(B, H, W, device=self.device) if want_filt else None
        _, h, w, _ = self.d2.shape
        _lib.call("noise_fused_tail", ptr(self.d2), ptr(self.a1), self.a1.shape[1], self.a1.shape[2], *self.crops[3],
                  ptr(self.a0), self.a0.shape[1], self.a0.shape[2], *self.crops[4], ptr(tw.wd), ptr(tw.b2), ptr(tw.w3),
                  ptr(tw.b3), ptr(tw.wl), tw.bias, ptr(logmag), ptr(absmax), ptr(self.enh), ptr(filt), B, h, w, H, W, st,
                  nbytes=2 * (self.d2.numel() + self.d2.numel() + 2 * self.d2.numel()) + 8 * B * H * W)
        return (self.enh, self.vad) if not want_filt else (self.enh, self.vad, filt)
