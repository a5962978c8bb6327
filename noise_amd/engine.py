"""Kernel schedule of the denoiser U-Net: forward, backward and the fused training step.

This is the host-side orchestration of the HIP kernels for `Base._forward` / `BaseSpectral._forward`
/ `_shared_step` of the reference (train/models.py:583-612, :758-764, :766-804).  It owns no
parameters: it reads them from the module tree built in `noise_amd.models` and writes gradients into
the tensors its caller designates (views of one flat gradient buffer).

Data layout: every activation is NHWC fp32 and stored RAW (before BatchNorm); consumers apply the
producer's BatchNorm(+ReLU6) while loading ("lazy" sources), and each conv kernel emits the
per-channel sums that the BatchNorm of its own output needs, so an activation is written once and
read once per consumer.
"""
import os

import torch

from . import _lib, ops
from ._lib import ptr
import ctypes

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_U64 = ctypes.c_ulonglong
_lib.register({
    "noise_attn_pre_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "noise_attn_core_fwd": [_P, _P, _P, _P, _U64, _F, _L, _I, _I, _I, _I, _P],
    "noise_attn_core_bwd": [_P, _P, _P, _P, _U64, _F, _P, _L, _I, _I, _I, _I, _P],
    "noise_attn_post_fwd": [_P] * 22 + [_L, _I, _I, _I, _I, _P],
    "noise_attn_post_bwd": [_P] * 20 + [_L, _I, _I, _I, _I, _P],
    "noise_attn_pre_bwd": [_P] * 12 + [_L, _I, _I, _I, _I, _P],
    "noise_spectral_l1": [_P, _P, _P, _P, _P, _L, _P],
    "noise_vad_label": [_P, _P, _I, _I, _I, _P],
    "noise_wave_pair_l1": [_P, _P, _P, _P, _P, _I, _I, _P],
    "noise_loss_finalize": [_P, _I, _L, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    "noise_adamw_step": [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P],
    "noise_absmax": [_P, _P, _I, _L, _P],
})
_lib.lib.noise_attn_max_blocks.restype = ctypes.c_int
_lib.lib.noise_loss_max_blocks.restype = ctypes.c_int
ATTN_MAX_BLOCKS = _lib.lib.noise_attn_max_blocks()
LOSS_MAX_BLOCKS = _lib.lib.noise_loss_max_blocks()

ATTN_DROPOUT = 0.1  # ScaledDotProductAttention default, never overridden by the model (models.py:22,55)


class _Rec:
    """One convolution layer of the schedule and the buffers that belong to it."""
    __slots__ = ("name", "kind", "conv", "bn", "relu6", "C", "Cin", "B", "Hi", "Wi", "Ho", "Wo", "stride",
                 "transposed", "y", "coef", "coef3", "src_a", "src_b", "dz", "g_in", "wslab", "virt")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    @property
    def npix_out(self):
        return self.B * self.Ho * self.Wo

    @property
    def npix_in(self):
        return self.B * self.Hi * self.Wi

    def lazy(self):
        return (self.y, self.coef, self.relu6)


class DenoiserPlan:
    """Buffers + launch schedule for one (batch, H, W) geometry of one VADMobileNetV2 instance."""

    def __init__(self, net, B, H, W, device):
        self.net = net
        self.B, self.H, self.W = B, H, W
        self.device = device
        self.hp, self.wp = net.H_padding, net.W_padding
        self.pinned = 0     # > 0 while a captured HIP graph addresses this plan's buffers
        f32 = dict(device=device, dtype=torch.float32)
        self.recs = []
        enc_blocks = net.block_plan()[0]
        dec_blocks = net.block_plan()[1]

        def new_rec(name, kind, conv, bn, relu6, Cin, C, Hi, Wi, stride, transposed):
            Ho = ops.conv_out(Hi, stride[0], transposed)
            Wo = ops.conv_out(Wi, stride[1], transposed)
            r = _Rec(name=name, kind=kind, conv=conv, bn=bn, relu6=relu6, C=C, Cin=Cin, B=B, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo,
                     stride=stride, transposed=transposed)
            r.y = torch.empty(B, Ho, Wo, C, **f32)
            r.coef = torch.empty(5 * C, **f32)
            r.coef3 = torch.empty(6 * C, **f32)
            self.recs.append(r)
            return r

        # ---- encoder
        self.enc = []       # list of lists of recs per block
        Hc, Wc = H + self.hp, W + self.wp
        blk = enc_blocks[0]
        m = net.encoder[0]
        r = new_rec("encoder.0", "first", m[0], m[1], True, 1, blk["cout"], Hc, Wc, blk["stride"], False)
        self.enc.append([r])
        for i, blk in enumerate(enc_blocks[1:], start=1):
            seq = net.encoder[i].conv
            layers = []
            idx = 0
            cur = self.enc[-1][-1]
            if blk["t"] != 1:
                layers.append(new_rec(f"encoder.{i}.pw", "pw", seq[0][0], seq[0][1], True, blk["cin"], blk["hidden"],
                                      cur.Ho, cur.Wo, (1, 1), False))
                idx = 1
                cur = layers[-1]
            layers.append(new_rec(f"encoder.{i}.dw", "dw", seq[idx][0], seq[idx][1], True, blk["hidden"], blk["hidden"],
                                  cur.Ho, cur.Wo, blk["stride"], False))
            cur = layers[-1]
            layers.append(new_rec(f"encoder.{i}.lin", "pw", seq[idx + 1], seq[idx + 2], False, blk["hidden"], blk["cout"],
                                  cur.Ho, cur.Wo, (1, 1), False))
            self.enc.append(layers)
        bott = self.enc[-1][-1]
        self.n_tok = bott.Ho * bott.Wo
        self.d_model = bott.C
        self.ntok = B * self.n_tok
        # ---- attention buffers
        hd = net.n_head * net.d_k
        self.hd = hd
        nt, D = self.ntok, self.d_model
        self.xhat0 = torch.empty(nt, D, **f32)
        self.rstd0 = torch.empty(nt, **f32)
        self.qkv = torch.empty(nt, 3 * hd, **f32)
        self.att_o = torch.empty(nt, hd, **f32)
        self.ml = torch.empty(B * net.n_head * self.n_tok * 2, **f32)
        self.xhat1 = torch.empty(nt, D, **f32)
        self.rstd1 = torch.empty(nt, **f32)
        self.hdn = torch.empty(nt, D, **f32)
        self.xhat2 = torch.empty(nt, D, **f32)
        self.rstd2 = torch.empty(nt, **f32)
        self.x3 = torch.empty(nt, D, **f32)
        self.vad = torch.empty(B, **f32)
        # ---- decoder
        self.dec = []
        self.crops = {}     # decoder block -> (cropped copy of the skip tensor, row offset, column offset)
        Hc, Wc = bott.Ho, bott.Wo
        for i, blk in enumerate(dec_blocks):
            skip = self.enc[len(self.enc) - 1 - i][-1]
            if skip.Ho < Hc or skip.Wo < Wc:
                raise RuntimeError(f"decoder block {i}: skip {skip.Ho}x{skip.Wo} is smaller than the decoder tensor "
                                   f"{Hc}x{Wc} (the reference's `x + skip[..., -h:, -w:]` fails to broadcast as well)")
            if (skip.Ho, skip.Wo) != (Hc, Wc):
                # the padding was searched for another input size: keep the last Hc rows / Wc columns of the skip tensor
                # (train/models.py:608)
                self.crops[i] = (torch.empty(B, Hc, Wc, skip.C, **f32), skip.Ho - Hc, skip.Wo - Wc)
            if blk["kind"] == "last":
                m = net.decoder[i]
                r = _Rec(name=f"decoder.{i}", kind="last", conv=m[0], bn=None, relu6=False, C=1, Cin=blk["cin"], B=B, Hi=Hc,
                         Wi=Wc, Ho=H, Wo=W, stride=blk["stride"], transposed=True)
                full_h, full_w = ops.conv_out(Hc, blk["stride"][0], True), ops.conv_out(Wc, blk["stride"][1], True)
                if full_h < H or full_w < W:
                    raise RuntimeError(f"U-Net output {full_h}x{full_w} is smaller than its input {H}x{W}: "
                                       "H_padding/W_padding do not fit this input size")
                self.recs.append(r)
                self.dec.append([r])
                continue
            seq = net.decoder[i].conv
            layers = []
            idx = 0
            if blk["t"] != 1:
                layers.append(new_rec(f"decoder.{i}.pw", "pw", seq[0][0], seq[0][1], True, blk["cin"], blk["hidden"], Hc, Wc,
                                      (1, 1), False))
                idx = 1
            layers.append(new_rec(f"decoder.{i}.dw", "dw", seq[idx][0], seq[idx][1], True, blk["hidden"], blk["hidden"], Hc,
                                  Wc, blk["stride"], True))
            cur = layers[-1]
            layers.append(new_rec(f"decoder.{i}.lin", "pw", seq[idx + 1], seq[idx + 2], False, blk["hidden"], blk["cout"],
                                  cur.Ho, cur.Wo, (1, 1), False))
            self.dec.append(layers)
            Hc, Wc = cur.Ho, cur.Wo
        self.enh = torch.empty(B, H, W, **f32)
        # ---- scratch shared by all layers
        maxC = max(r.C for r in self.recs)
        maxw = max((r.C * r.Cin * ops.pw_bwd_max_rows(r.C, r.Cin) for r in self.recs if r.kind == "pw"), default=1)
        att_cols = max(2 * D * D + D * hd + 7 * D + 1, 3 * hd * D + 2 * D)
        self.slab = torch.empty(max(ops.CONV_MAX_BLOCKS * 12 * maxC, maxw, ATTN_MAX_BLOCKS * att_cols), **f32)
        self.slab2 = torch.empty(1024 * 2 * maxC, **f32)
        self.att_grads = torch.empty(att_cols, **f32)
        # per-workgroup weight-gradient rows of the two attention backward kernels (folded with the other layers')
        self.att_slab_post = torch.empty(ATTN_MAX_BLOCKS * (2 * D * D + D * hd + 7 * D + 1), **f32)
        self.att_slab_pre = torch.empty(ATTN_MAX_BLOCKS * (3 * hd * D + 2 * D), **f32)
        self.bwd_ready = False
        # the output of the decoder's stride-2 transposed depthwise layers is never materialised (NOISE_VIRTUAL_DW=0: it is)
        # (a comma-separated list of channel counts restricts it to those layers)
        v = os.environ.get("NOISE_VIRTUAL_DW", "1")
        self.virtual_dw = v != "0" and "NOISE_DW_NO_RECOMPUTE" not in os.environ
        self.virtual_dw_channels = {int(c) for c in v.split(",")} if "," in v or v not in ("0", "1") else None

    # -------------------------------------------------------------------------------- backward buffers
    def ensure_backward_buffers(self):
        if self.bwd_ready:
            return
        f32 = dict(device=self.device, dtype=torch.float32)
        # a cropped skip (train/models.py:608 with a padding searched for another length): the gradient of the window
        # goes back to the encoder tensor zero-padded to its full size
        self.crop_grads = {}
        for i, (win, oy, ox) in self.crops.items():
            skip = self.enc[len(self.enc) - 1 - i][-1]
            self.crop_grads[i] = torch.empty(skip.B, skip.Ho, skip.Wo, skip.C, **f32)
        for r in self.recs:
            if r.kind != "first":
                r.g_in = torch.empty(r.B, r.Hi, r.Wi, r.Cin, **f32)
            # every inner layer keeps its own weight-gradient slab: nothing waits for a weight gradient before the
            # optimizer, so the slabs are folded by ONE launch at the end of the backward pass (see _fold_weight_grads)
            if r.kind == "pw":
                r.wslab = torch.empty(ops.pw_bwd_max_rows(r.C, r.Cin) * r.C * r.Cin, **f32)
            elif r.kind == "dw":
                r.wslab = torch.empty(ops.CONV_MAX_BLOCKS * 11 * r.C, **f32)
        self.fold_table = None      # (key, device job table, njobs, max_cols)
        nt, D, hd = self.ntok, self.d_model, self.hd
        self.d_o = torch.empty(nt, hd, **f32)
        self.d_x1 = torch.empty(nt, D, **f32)
        self.d_qkv = torch.empty(nt, 3 * hd, **f32)
        self.d_x0 = torch.empty(nt, D, **f32)
        self.bwd_ready = True

    # -------------------------------------------------------------------------------- forward
    def _bn(self, r, rows, training):
        bn = r.bn
        if training:
            ops.bn_finalize_fwd(self.slab, rows, r.C, r.npix_out, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                bn.num_batches_tracked, r.coef, momentum=bn.momentum, eps=bn.eps)
        else:
            ops.bn_coef_eval(r.C, bn.weight, bn.bias, bn.running_mean, bn.running_var, r.coef, eps=bn.eps)

    def _run_layer(self, r, a, b, training):
        slab = self.slab if training else None
        r.src_a, r.src_b = a, b
        if r.kind == "pw":
            rows = ops.pw_fwd(a, b, r.conv.weight, r.y, slab, r.npix_in, r.Cin, r.C)
        else:
            rows = ops.dw3x3_fwd(a, b, r.conv.weight, r.y, slab, r.transposed, r.B, r.Hi, r.Wi, r.C, r.stride)
        self._bn(r, rows, training)

    def forward(self, logmag, absmax, training, keep=None, seed=0):
        """logmag (B,H,W) raw log-magnitudes, absmax (B,) -> enhanced log-magnitudes (B,H,W), vad (B,)."""
        net = self.net
        self.logmag, self.absmax = logmag, absmax
        if not training:
            keep = None     # eval / validation forwards never drop attention weights
        self.keep, self.seed = keep, int(seed)
        self.p_drop = ATTN_DROPOUT if training else 0.0
        if training:
            net.weights_changed()      # the finalize kernels update the BatchNorm running statistics in place
        r0 = self.enc[0][0]
        rows = ops.first_conv_fwd(logmag, absmax, r0.conv.weight, r0.y, self.slab if training else None, self.B, self.H,
                                  self.W, self.hp, self.wp, net.pad_value, r0.C, r0.stride)
        self._bn(r0, rows, training)
        cur = r0.lazy()
        for layers in self.enc[1:]:
            for r in layers:
                self._run_layer(r, cur, None, training)
                cur = r.lazy()
        bott = cur
        # ---- attention block
        att = net.bottleneck_attention
        ly = att.layer_stack[0]
        sa, ff = ly.slf_attn, ly.pos_ffn
        st = _lib.stream()
        dims = (self.ntok, self.n_tok, self.d_model, net.n_head, net.d_k)
        _lib.call("noise_attn_pre_fwd", ptr(bott[0]), ptr(bott[1]), int(bott[2]), ptr(att.layer_norm.weight),
                  ptr(att.layer_norm.bias), ptr(sa.w_qs.weight), ptr(sa.w_ks.weight), ptr(sa.w_vs.weight), ptr(self.xhat0),
                  ptr(self.rstd0), ptr(self.qkv), *dims, st, nbytes=4 * self.ntok * (2 * self.d_model + 3 * self.hd))
        _lib.call("noise_attn_core_fwd", ptr(self.qkv), ptr(self.att_o), ptr(self.ml), ptr(keep), self.seed,
                  float(self.p_drop if keep is None else ATTN_DROPOUT), *dims, st, nbytes=4 * self.ntok * 4 * self.hd)
        _lib.call("noise_attn_post_fwd", ptr(self.att_o), ptr(self.xhat0), ptr(att.layer_norm.weight),
                  ptr(att.layer_norm.bias), ptr(sa.fc.weight), ptr(sa.layer_norm.weight), ptr(sa.layer_norm.bias),
                  ptr(ff.w_1.weight), ptr(ff.w_1.bias), ptr(ff.w_2.weight), ptr(ff.w_2.bias), ptr(ff.layer_norm.weight),
                  ptr(ff.layer_norm.bias), ptr(net.vad.weight), ptr(net.vad.bias), ptr(self.xhat1), ptr(self.rstd1),
                  ptr(self.hdn), ptr(self.xhat2), ptr(self.rstd2), ptr(self.x3), ptr(self.vad), *dims, st,
                  nbytes=4 * self.ntok * (self.hd + 5 * self.d_model))
        # ---- decoder
        cur = (self.x3, None, False)
        for i, layers in enumerate(self.dec):
            skip = self.enc[len(self.enc) - 1 - i][-1].lazy()
            if i in self.crops:
                win, oy, ox = self.crops[i]
                ops.crop_nhwc(skip[0], win, oy, ox)
                skip = (win, skip[1], skip[2])
            first = True
            virt = None
            for r in layers:
                if virt is not None:
                    # projection of a transposed block: the depthwise output is rebuilt inside the 1x1 convolution
                    dw, virt = virt, None
                    r.src_a, r.src_b, r.virt = dw.lazy(), None, dw
                    rows = ops.pw_fwd_virt(dw.src_a, dw.conv.weight, dw.coef, dw.relu6, r.conv.weight, r.y,
                                           self.slab if training else None, dw.B, dw.Hi, dw.Wi, r.Cin, r.C)
                    self._bn(r, rows, training)
                    cur = r.lazy()
                    continue
                r.virt = None
                if (self.virtual_dw and not first and r.kind == "dw" and r.transposed and tuple(r.stride) == (2, 2)
                        and (self.virtual_dw_channels is None or r.C in self.virtual_dw_channels)
                        and ops.pw_fwd_virt_supported(r.C, layers[-1].C)):
                    # nothing reads the depthwise output but the projection behind it (forward and backward) and this
                    # layer's own backward, which rebuilds it as well: a training forward only takes its statistics
                    r.src_a, r.src_b = cur, None
                    rows = 0
                    if training:
                        rows = ops.dw3x3_fwd(cur, None, r.conv.weight, None, self.slab, True, r.B, r.Hi, r.Wi, r.C, r.stride)
                    self._bn(r, rows, training)
                    virt = r
                    continue
                if r.kind == "last":
                    r.src_a, r.src_b = cur, skip
                    ops.last_convT_fwd(cur, skip, r.conv.weight, r.conv.bias, logmag, absmax, self.enh, None, r.B, r.Hi,
                                       r.Wi, r.Cin, r.stride, self.H, self.W)
                else:
                    self._run_layer(r, cur, skip if first else None, training)
                    cur = r.lazy()
                first = False
        return self.enh, self.vad

    # -------------------------------------------------------------------------------- backward
    def backward(self, d_enh, d_vad, grad_of):
        """d_enh (B,H,W) = dL/d(enhanced log-magnitude), d_vad (B,) = dL/d(vad logit).
        `grad_of(param)` returns the tensor that receives that parameter's gradient (overwritten).

        Every layer's backward kernel also finishes the BatchNorm backward of the layer that PRODUCED its input:
        it adds the pending skip-branch gradient, applies the producer's ReLU6 mask and leaves the two per-channel
        sums, so the only work left per BatchNorm is a tiny finalize kernel."""
        self.ensure_backward_buffers()
        net = self.net
        G = grad_of
        by_y = {id(r.y): r for r in self.recs if r.y is not None}
        pending_skip = {}      # id(rec) -> gradient w.r.t. that rec's activation coming from a decoder skip add

        # cropped copies of skip tensors stand for the encoder layer they were cut from
        crop_of = {id(win): (self.enc[len(self.enc) - 1 - i][-1], self.crop_grads[i], oy, ox)
                   for i, (win, oy, ox) in self.crops.items()}

        def producer(src):
            if src is None:
                return None
            if id(src[0]) in crop_of:
                return crop_of[id(src[0])][0]
            return by_y.get(id(src[0]))

        def epilogue_for(pa):
            if pa is None:
                return None, None
            return pending_skip.pop(id(pa), None), (pa.coef, pa.relu6)

        def finalize(pa, slab_view, rows, row_stride):
            if pa is not None:
                ops.bn_bwd_finalize(slab_view, rows, row_stride, pa.coef, pa.bn.weight, G(pa.bn.weight), G(pa.bn.bias),
                                    pa.coef3, pa.npix_out, pa.C)

        def done(r, pa, pb):
            if pa is not None:
                pa.dz = r.g_in
            if pb is not None:
                g = r.g_in
                if r.src_b is not None and id(r.src_b[0]) in crop_of:
                    _, g, oy, ox = crop_of[id(r.src_b[0])]
                    ops.crop_nhwc_bwd(r.g_in, g, oy, ox)
                pending_skip[id(pb)] = g

        # ---- last layer
        last = self.dec[-1][0]
        pa, pb = producer(last.src_a), producer(last.src_b)
        g_other, epi = epilogue_for(pa)
        C = last.Cin
        rows = ops.last_convT_bwd(d_enh, self.enh, self.absmax, last.src_a, last.src_b, last.conv.weight, last.g_in,
                                  G(last.conv.weight), G(last.conv.bias), self.slab, last.B, last.Hi, last.Wi, C,
                                  last.stride, self.H, self.W, g_other, epi)
        finalize(pa, self.slab[10 * C:], rows, 12 * C)
        done(last, pa, pb)

        def layer_backward(r):
            grad = (r.dz, r.y, r.coef3)
            pa, pb = producer(r.src_a), producer(r.src_b)
            g_other, epi = epilogue_for(pa)
            if r.kind == "pw":
                if r.virt is not None:
                    dw = r.virt     # == pa: the depthwise layer whose output was never written
                    rows = ops.pw_bwd_virt(grad, dw.src_a, dw.conv.weight, dw.coef, dw.relu6, r.conv.weight, r.g_in,
                                           r.wslab, self.slab2, dw.B, dw.Hi, dw.Wi, r.Cin, r.C)
                else:
                    rows = ops.pw_bwd(grad, r.src_a, r.src_b, r.conv.weight, r.g_in, None, r.wslab, r.npix_in,
                                      r.C, r.Cin, g_other, epi, self.slab2)
                finalize(pa, self.slab2, rows, 2 * r.Cin)
                fold.append((r.wslab, G(r.conv.weight), rows, r.C * r.Cin, r.C * r.Cin, 0, 0))
            else:
                rows = ops.dw3x3_bwd(grad, r.src_a, r.src_b, r.conv.weight, r.g_in, None, r.wslab,
                                     r.transposed, r.B, r.Hi, r.Wi, r.C, r.stride, g_other, epi)
                finalize(pa, r.wslab[9 * r.C:], rows, 11 * r.C)
                fold.append((r.wslab, G(r.conv.weight), rows, 9 * r.C, 11 * r.C, 9, r.C))
            done(r, pa, pb)

        fold = []   # (slab, destination, rows, columns, row stride, tr_k, tr_C) of every deferred weight gradient
        for layers in reversed(self.dec[:-1]):
            for r in reversed(layers):
                layer_backward(r)
        # ---- attention block: dL/dx3 is the input gradient of the first decoder layer
        g3 = self.dec[0][0].g_in
        att = net.bottleneck_attention
        ly = att.layer_stack[0]
        sa, ff = ly.slf_attn, ly.pos_ffn
        st = _lib.stream()
        dims = (self.ntok, self.n_tok, self.d_model, net.n_head, net.d_k)
        D, hd = self.d_model, self.hd
        att_rows = min(-(-self.ntok // 64), ATTN_MAX_BLOCKS)
        post_stride, pre_stride = 2 * D * D + D * hd + 7 * D + 1, 3 * hd * D + 2 * D
        _lib.call("noise_attn_post_bwd", ptr(g3), ptr(d_vad), ptr(self.att_o), ptr(self.xhat1), ptr(self.rstd1),
                  ptr(self.hdn), ptr(self.xhat2), ptr(self.rstd2), ptr(self.x3), ptr(sa.fc.weight),
                  ptr(sa.layer_norm.weight), ptr(sa.layer_norm.bias), ptr(ff.w_1.weight), ptr(ff.w_2.weight),
                  ptr(ff.layer_norm.weight), ptr(net.vad.weight), ptr(self.d_o), ptr(self.d_x1), None,
                  ptr(self.att_slab_post), *dims, st, nbytes=4 * self.ntok * (2 * self.hd + 7 * self.d_model))
        o = 0
        for t, n in ((ff.w_2.weight, D * D), (ff.w_1.weight, D * D), (sa.fc.weight, D * hd), (ff.layer_norm.weight, D),
                     (ff.layer_norm.bias, D), (ff.w_2.bias, D), (ff.w_1.bias, D), (sa.layer_norm.weight, D),
                     (sa.layer_norm.bias, D), (net.vad.weight, D), (net.vad.bias, 1)):
            fold.append((self.att_slab_post[o:], G(t).view(-1), att_rows, n, post_stride, 0, 0))
            o += n
        _lib.call("noise_attn_core_bwd", ptr(self.qkv), ptr(self.d_o), ptr(self.ml), ptr(self.keep), self.seed,
                  float(self.p_drop if self.keep is None else ATTN_DROPOUT), ptr(self.d_qkv), *dims, st,
                  nbytes=4 * self.ntok * 7 * self.hd)
        _lib.call("noise_attn_pre_bwd", ptr(self.d_qkv), ptr(self.d_x1), ptr(self.xhat0), ptr(self.rstd0),
                  ptr(att.layer_norm.weight), ptr(att.layer_norm.bias), ptr(sa.w_qs.weight), ptr(sa.w_ks.weight),
                  ptr(sa.w_vs.weight), ptr(self.d_x0), None, ptr(self.att_slab_pre), *dims, st,
                  nbytes=4 * self.ntok * (3 * self.hd + 3 * self.d_model))
        o = 0
        for t, n in ((sa.w_qs.weight, hd * D), (sa.w_ks.weight, hd * D), (sa.w_vs.weight, hd * D),
                     (att.layer_norm.weight, D), (att.layer_norm.bias, D)):
            fold.append((self.att_slab_pre[o:], G(t).view(-1), att_rows, n, pre_stride, 0, 0))
            o += n
        # the bottleneck tensor gets dL/dx0 from the attention block plus the skip gradient of decoder.0: this one
        # BatchNorm backward keeps its own (small) reduction pass
        bott = self.enc[-1][-1]
        d_x0 = self.d_x0.view(bott.B, bott.Ho, bott.Wo, bott.C)
        skip = pending_skip.pop(id(bott), None)
        ops.bn_bwd(d_x0, skip, bott.y, bott.coef, bott.relu6, bott.bn.weight, d_x0 if (bott.relu6 or skip is not None)
                   else None, G(bott.bn.weight), G(bott.bn.bias), bott.coef3, self.slab, bott.npix_out, bott.C)
        bott.dz = d_x0
        # ---- encoder
        for layers in reversed(self.enc[1:]):
            for r in reversed(layers):
                layer_backward(r)
        r0 = self.enc[0][0]
        ops.first_conv_wgrad((r0.dz, r0.y, r0.coef3), self.logmag, self.absmax, G(r0.conv.weight), self.slab, self.B,
                             self.H, self.W, self.hp, self.wp, net.pad_value, r0.C, r0.stride, w=r0.conv.weight)
        assert not pending_skip, "unconsumed skip gradients"
        self._fold_weight_grads(fold)

    def _fold_weight_grads(self, fold):
        """One launch folds all deferred weight-gradient slabs.  The job table lives on the device and is rebuilt only
        when a pointer or row count changes (a new gradient target, another geometry)."""
        key = tuple((sl.data_ptr(), out.data_ptr(), rows) for sl, out, rows, *_ in fold)
        if self.fold_table is None or self.fold_table[0] != key:
            jobs = (ops.FoldJob * len(fold))()
            for j, (sl, out, rows, ncols, stride, tr_k, tr_C) in enumerate(fold):
                jobs[j] = ops.FoldJob(sl.data_ptr(), out.data_ptr(), rows, ncols, stride, tr_k, tr_C)
            host = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
            self.fold_table = (key, host.to(self.device), len(fold), max(f[3] for f in fold))
        _, table, njobs, max_cols = self.fold_table
        _lib.call("noise_fold_jobs", ptr(table), njobs, max_cols, _lib.stream())
