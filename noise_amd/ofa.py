"""The once-for-all spectral U-Net of train/layers.py (`OFASpectralNets`, :496-668) on the shape-generic HIP kernels.

The module tree, parameter names and shapes are the reference's (1000 state_dict entries with the default arguments, built
in the reference's order, so a model made under the same seed has the reference's weights and its checkpoints load), and
so are the run-time knobs: `runtime_depth` per stage, `active_kernel_size` / `active_expand_ratio` / `active_out_channel`
per block.  The `nn.Conv2d` / `nn.BatchNorm2d` objects are parameter containers; the forward pass runs on the GPU through
`noise_ofa_*` (csrc/ofa.hip): activations are NHWC fp32, a convolution writes its raw output with partial BatchNorm sums,
the next kernel applies BatchNorm + ReLU6 while loading, and one elementwise kernel per block materialises the block
output with the residual input (layers.py:457-458) and the cropped encoder skip (layers.py:662) added.  Train mode uses
batch statistics and updates the running ones (DynamicBatchNorm2d on the leading channels, layers.py:284-308), eval mode
uses the running statistics.  Forward only: the reference never trains this class (nothing in train/main.py reaches it).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ptr

_P, _I, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
_lib.register({
    "noise_ofa_first_conv": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "noise_ofa_dw_filter": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P],
    "noise_ofa_dw": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "noise_ofa_pw": [_P, _P, _I, _P, _I, _P, _P, _P, _L, _I, _I, _P],
    "noise_ofa_apply": [_P, _P, _I, _I, _I, _P, _P, _I, _I, _P, _I, _I, _I, _I, _P],
    "noise_ofa_last_convT": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "noise_ofa_final": [_P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
})
_lib.lib.noise_ofa_max_rows.restype = ctypes.c_int
_lib.lib.noise_ofa_max_rows.argtypes = []

STAGE_WIDTHS = {   # layers.py:525-531
    "google": [32, 16, 24, 32, 64, 96, 160, 320, 1280],
    "proxyless": [32, 16, 24, 40, 80, 96, 192, 320, 1280],
    "mcunet384": [32, 16, 24, 40, 80, 96, 192, 320, 384],
}


def make_divisible(v, divisor, min_val=None):
    """Nearest multiple of `divisor` (at least `min_val`), never more than 10 % below v (layers.py:57-74)."""
    floor = divisor if min_val is None else min_val
    r = max(floor, (int(v + divisor / 2) // divisor) * divisor)
    return r + divisor if r < 0.9 * v else r


def _as_list(v, n=1):
    if isinstance(v, (list, tuple)):
        return list(v)
    return [v] * n


# ---------------------------------------------------------------------------------------------- parameter containers
class DynamicPointConv2d(nn.Module):
    """layers.py:218-272: a dense convolution whose leading output / input channels are the active ones."""

    def __init__(self, max_in_channels, max_out_channels, kernel_size=1, stride=1, conv_layer=nn.Conv2d):
        super().__init__()
        self.max_in_channels, self.max_out_channels = max_in_channels, max_out_channels
        self.kernel_size, self.stride = kernel_size, stride
        self.transposed = conv_layer is nn.ConvTranspose2d
        self.conv = conv_layer(max_in_channels, max_out_channels, kernel_size, stride=stride, bias=False)
        self.active_out_channel = max_out_channels


class DynamicBatchNorm2d(nn.Module):
    """layers.py:275-313: BatchNorm over the leading `feature_dim` channels of its parameters and buffers."""

    def __init__(self, max_feature_dim):
        super().__init__()
        self.max_feature_dim = max_feature_dim
        self.bn = nn.BatchNorm2d(max_feature_dim)


class DynamicConv2dNormActivation(nn.Module):
    """layers.py:316-360."""

    def __init__(self, in_channel_list, out_channel_list, kernel_size=3, stride=1, use_bn=True, activation="relu6",
                 conv_layer=nn.Conv2d):
        super().__init__()
        self.in_channel_list, self.out_channel_list = in_channel_list, out_channel_list
        self.kernel_size, self.stride, self.use_bn, self.activation = kernel_size, stride, use_bn, activation
        self.conv = DynamicPointConv2d(max(in_channel_list), max(out_channel_list), kernel_size, stride, conv_layer)
        if use_bn:
            self.bn = DynamicBatchNorm2d(max(out_channel_list))
        self.active_out_channel = max(out_channel_list)


class DynamicSeparableConv2d(nn.Module):
    """layers.py:108-215: depthwise filter of the largest kernel size; smaller sizes are its centre passed through the
    `<larger>to<smaller>_matrix` parameters (identity at construction)."""

    def __init__(self, max_in_channels, kernel_size_list, stride=1, conv_layer=nn.Conv2d):
        super().__init__()
        self.max_in_channels, self.kernel_size_list, self.stride = max_in_channels, kernel_size_list, stride
        self.transposed = conv_layer is nn.ConvTranspose2d
        kmax = max(kernel_size_list)
        self.conv = conv_layer(max_in_channels, max_in_channels, kmax, stride=stride, groups=max_in_channels, bias=False)
        self._ks_set = sorted(set(kernel_size_list))
        for small, large in zip(self._ks_set[:-1], self._ks_set[1:]):
            self.register_parameter(f"{large}to{small}_matrix", nn.Parameter(torch.eye(small * small)))
        self.active_kernel_size = kmax

    def transform_steps(self, kernel_size):
        """[(matrix, target size)] from the largest size down to `kernel_size` (layers.py:172-194)."""
        steps = []
        for small, large in reversed(list(zip(self._ks_set[:-1], self._ks_set[1:]))):
            if large <= kernel_size:
                break
            steps.append((getattr(self, f"{large}to{small}_matrix"), small))
        return steps


class _DepthConv(nn.Module):
    """The `depth_conv` Sequential of layers.py:407-424 (conv, bn; its activation has no state)."""

    def __init__(self, channels, kernel_size_list, stride, conv_layer):
        super().__init__()
        self.conv = DynamicSeparableConv2d(channels, kernel_size_list, stride, conv_layer)
        self.bn = DynamicBatchNorm2d(channels)


class DynamicInvertedResidual(nn.Module):
    """layers.py:363-460: expand 1x1 (absent at ratio 1) -> depthwise k x k -> linear 1x1, residual when the stride is
    1 and the widths agree."""

    def __init__(self, in_channel_list, out_channel_list, kernel_size_list, expand_ratio_list, stride=1,
                 activation="relu6", dw_layer=None):
        super().__init__()
        self.in_channel_list, self.out_channel_list = in_channel_list, out_channel_list
        self.kernel_size_list, self.expand_ratio_list = _as_list(kernel_size_list), _as_list(expand_ratio_list)
        self.stride, self.activation = stride, activation
        middle = round(max(in_channel_list) * max(self.expand_ratio_list))
        self.active_kernel_size = max(self.kernel_size_list)
        self.active_expand_ratio = max(self.expand_ratio_list)
        self.active_out_channel = max(out_channel_list)
        self.inverted_bottleneck = None
        self.depth_conv = None
        if max(self.expand_ratio_list) > 1:
            self.inverted_bottleneck = DynamicConv2dNormActivation(_as_list(max(in_channel_list)), _as_list(middle), 1, 1,
                                                                   True, activation)
        if dw_layer is not None:
            self.depth_conv = _DepthConv(middle, self.kernel_size_list, stride, dw_layer)
        self.point_linear = DynamicConv2dNormActivation(_as_list(middle), _as_list(max(out_channel_list)), 1, 1, True, None)


# ------------------------------------------------------------------------------------------------------------ network
class OFASpectralNets(nn.Module):
    """Constructor arguments, attributes and forward semantics of layers.py:496-668."""

    def __init__(self, in_channels=1, out_channels=1, base_stage_width="mcunet384", base_depth=(1, 2, 2, 2, 2, 2, 2),
                 depth_list=(0, 1, 2), width_mult_list=(0.5, 0.75, 1.0), kernel_size_list=(3, 5, 7),
                 expand_ratio_list=(3, 4, 6), stride_list=(1, 2, 2, 2, 1, 2, 1), activation="relu6"):
        super().__init__()
        if in_channels != 1 or out_channels != 1:
            raise NotImplementedError("the first / last layer kernels take one spectrogram channel (the reference's default)")
        if activation != "relu6":
            raise NotImplementedError("only the reference's default activation (relu6) has a kernel")
        self.width_mult_list = sorted(_as_list(width_mult_list))
        self.kernel_size_list = sorted(_as_list(kernel_size_list))
        self.expand_ratio_list = sorted(_as_list(expand_ratio_list))
        self.depth_list = sorted(_as_list(depth_list))
        if any(k % 2 == 0 or k > 7 for k in self.kernel_size_list) or len(set(self.kernel_size_list)) > 3:
            raise NotImplementedError("depthwise kernel sizes: at most three odd sizes up to 7")
        self.base_stage_width, self.base_depth = base_stage_width, list(base_depth)
        self.stride_list, self.activation = list(stride_list), activation
        widths = STAGE_WIDTHS.get(base_stage_width, base_stage_width) if isinstance(base_stage_width, str) else base_stage_width
        widths = list(widths)

        cin = [make_divisible(widths[0] * m, 8) for m in self.width_mult_list]
        self.block_group_info = [[0]]
        self.encoder = nn.ModuleList([DynamicConv2dNormActivation(_as_list(in_channels, len(cin)), cin, 3, 2, True, activation,
                                                                  nn.Conv2d)])
        self.decoder = nn.ModuleList([DynamicConv2dNormActivation(cin, _as_list(out_channels, len(cin)), 3, 2, True, activation,
                                                                  nn.ConvTranspose2d)])
        depths = [d + max(self.depth_list) * (i > 0) for i, d in enumerate(self.base_depth)]
        stage_widths = [[make_divisible(w * m, 8) for m in self.width_mult_list] for w in widths[1:]]
        index = 1
        ks, er = list(kernel_size_list), list(expand_ratio_list)
        for i, (cout, depth, stride) in enumerate(zip(stage_widths, depths, self.stride_list)):
            self.block_group_info.append(list(range(index, index + depth)))
            index += depth
            for j in range(depth):
                s = stride if j == 0 else 1
                self.encoder.append(DynamicInvertedResidual(cin, cout, ks, er if i != 0 else 1, s, "relu6", nn.Conv2d))
                self.decoder.insert(0, DynamicInvertedResidual(cout, cin, ks, er if i != 0 else 1, s, "relu6",
                                                               nn.ConvTranspose2d))
                cin = cout
        self.runtime_depth = [len(g) for g in self.block_group_info]
        self.H_padding = self.W_padding = 0
        self.searching = self.found = False

    # ------------------------------------------------------------------------------------------------ sub-network choice
    def set_active_subnet(self, wid=None, ks=None, e=None, d=None, **kwargs):
        """layers.py:723-748 as it behaves: the width loop runs, then the method fails on `self.blocks`, an attribute the
        reference class never defines.  `set_active` is the working form."""
        for m in self.modules():
            if hasattr(m, "out_channel_list"):
                m.active_out_channel = m.out_channel_list[wid] if wid is not None else max(m.out_channel_list)
        raise AttributeError("'OFASpectralNets' object has no attribute 'blocks'")

    def set_active(self, ks=None, e=None, d=None):
        """Kernel size and expand ratio of every inverted-residual block (both halves), and the depth of every stage."""
        for m in list(self.encoder[1:]) + list(self.decoder[:-1]):
            if ks is not None:
                m.active_kernel_size = ks
            if e is not None:
                m.active_expand_ratio = e
        if d is not None:
            self.runtime_depth = [min(len(g), dd) for g, dd in zip(self.block_group_info, _as_list(d, len(self.block_group_info)))]

    # ------------------------------------------------------------------------------------------------------- forward
    @torch.no_grad()
    def _search_padding(self, x):
        """layers.py:627-638: grow the top / left padding until the output covers the input.  Like the reference the search
        runs whole forwards (in train mode each of them updates the BatchNorm running statistics) and never clears
        `searching`, so it happens once per model."""
        self.searching = True
        H, W = x.shape[-2:]
        out = self.forward(x)
        for axis, name in ((-2, "H_padding"), (-1, "W_padding")):
            want = H if axis == -2 else W
            while out.shape[axis] < want:
                if getattr(self, name) > 256:
                    raise RuntimeError("OFASpectralNets: no padding makes the output cover the input with this sub-network "
                                       "(the active decoder blocks do not undo the encoder's strides)")
                setattr(self, name, getattr(self, name) + 1)
                out = self.forward(x)
        self.found = True

    def forward(self, x):
        _lib.require_cuda(x)
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"expected (B, 1, H, W) spectra, got {tuple(x.shape)}")
        if not self.searching and not self.found:
            self._search_padding(x)
        with torch.no_grad():
            return _Run(self, x).output()


class _Run:
    """One forward pass: owns the workspaces and walks the active blocks."""

    def __init__(self, net, x):
        self.net, self.dev = net, x.device
        self.B, _, self.H, self.W = x.shape
        self.x = x.reshape(self.B, self.H, self.W).contiguous().float()
        self.max_rows = _lib.lib.noise_ofa_max_rows()
        cmax = max(m.max_feature_dim for m in net.modules() if isinstance(m, DynamicBatchNorm2d))
        self.slab = torch.empty(self.max_rows * 2 * cmax, device=self.dev, dtype=torch.float32)
        self.rows = ctypes.c_int(0)

    def new(self, *shape):
        return torch.empty(*shape, device=self.dev, dtype=torch.float32)

    # -- BatchNorm over the leading C channels: slab of partial sums -> [scale | shift | ...] ------------------------
    def bn(self, dbn, C, count):
        bn = dbn.bn
        if C > bn.num_features:
            raise ValueError(f"{C} active channels exceed the BatchNorm's {bn.num_features}")
        coef = self.new(5 * C)
        if bn.training or not bn.track_running_stats:
            momentum = bn.momentum
            track = bn.training and bn.track_running_stats
            if track and momentum is None:     # cumulative moving average (layers.py:295-296)
                momentum = 1.0 / float(int(bn.num_batches_tracked) + 1)
            ops.bn_finalize_fwd(self.slab, self.rows.value, C, count, bn.weight, bn.bias,
                                bn.running_mean if track else None, bn.running_var if track else None,
                                bn.num_batches_tracked if track else None, coef, momentum=momentum or 0.0, eps=bn.eps)
        else:
            ops.bn_coef_eval(C, bn.weight, bn.bias, bn.running_mean, bn.running_var, coef, eps=bn.eps)
        return coef

    def pw(self, src, layer, K, N, npix):
        """DynamicConv2dNormActivation with a 1x1 kernel on the lazily normalised `src` -> (raw, coef)."""
        w = layer.conv.conv.weight
        if N > w.shape[0] or K > w.shape[1]:
            raise ValueError(f"active widths {K} -> {N} exceed the layer's {w.shape[1]} -> {w.shape[0]}")
        y, coef, relu6 = src
        out = self.new(npix, N)
        _lib.call("noise_ofa_pw", ptr(y), ptr(coef), int(relu6), ptr(w), w.shape[1], ptr(out), ptr(self.slab),
                  ctypes.byref(self.rows), npix, K, N, _lib.stream(), nbytes=4 * npix * (K + N))
        return out, self.bn(layer.bn, N, npix)

    def dw(self, src, dc, C, k, B, Hi, Wi):
        conv = dc.conv
        kmax = conv.conv.weight.shape[-1]
        if k not in conv._ks_set:
            raise ValueError(f"kernel size {k} is not one of {conv._ks_set}")
        if C > conv.conv.weight.shape[0]:
            raise ValueError(f"{C} active channels exceed the depthwise layer's {conv.conv.weight.shape[0]}")
        steps = conv.transform_steps(k) + [(None, 0), (None, 0)]
        (m1, k1), (m2, k2) = steps[0], steps[1]
        wt = self.new(k * k, C)
        s, tr = conv.stride, conv.transposed
        _lib.call("noise_ofa_dw_filter", ptr(conv.conv.weight), kmax, ptr(m1), k1, ptr(m2), k2, k, C, int(tr and s == 1),
                  ptr(wt), _lib.stream())
        pad = k // 2
        Ho = (Hi - 1) * s - 2 * pad + k if tr else (Hi + 2 * pad - k) // s + 1
        Wo = (Wi - 1) * s - 2 * pad + k if tr else (Wi + 2 * pad - k) // s + 1
        y, coef, relu6 = src
        out = self.new(B, Ho, Wo, C)
        _lib.call("noise_ofa_dw", ptr(y), ptr(coef), int(relu6), ptr(wt), ptr(out), ptr(self.slab), ctypes.byref(self.rows),
                  B, Hi, Wi, C, k, s, int(tr), _lib.stream(), nbytes=4 * C * B * (Hi * Wi + Ho * Wo))
        return out, self.bn(dc.bn, C, B * Ho * Wo), Ho, Wo

    def apply(self, y, coef, relu6, res=None, skip=None, out_hw=None):
        B, Hy, Wy, C = y.shape
        Ho, Wo = out_hw if out_hw is not None else (Hy, Wy)
        Hs, Ws = (skip.shape[1], skip.shape[2]) if skip is not None else (0, 0)
        if skip is not None and skip.shape[3] != C:
            raise RuntimeError(f"decoder tensor has {C} channels, the encoder skip {skip.shape[3]}")
        out = self.new(B, Ho, Wo, C)
        _lib.call("noise_ofa_apply", ptr(y), ptr(coef), int(relu6), Hy, Wy, ptr(res), ptr(skip), Hs, Ws, ptr(out), B, Ho, Wo,
                  C, _lib.stream(), nbytes=4 * out.numel() * (2 + (res is not None) + (skip is not None)))
        return out

    def block(self, blk, x, skip=None):
        """DynamicInvertedResidual.forward (layers.py:436-460) on a materialised NHWC tensor; `skip` is the encoder tensor
        the next decoder stage adds (layers.py:662), folded into this block's output pass."""
        B, Hi, Wi, Cin = x.shape
        src, C = (x, None, False), Cin
        if blk.inverted_bottleneck is not None:
            ib = blk.inverted_bottleneck
            # the block asks for make_divisible(Cin * ratio) channels (layers.py:442-444), then the layer's own forward
            # overwrites that with its `active_out_channel` (layers.py:353): the latter is what runs
            ib.conv.active_out_channel = make_divisible(round(Cin * blk.active_expand_ratio), 8)
            ib.conv.active_out_channel = ib.active_out_channel
            N = ib.conv.active_out_channel
            y, coef = self.pw(src, ib, C, N, B * Hi * Wi)
            src, C = (y.view(B, Hi, Wi, N), coef, True), N
        Ho, Wo = Hi, Wi
        if blk.depth_conv is not None:
            blk.depth_conv.conv.active_kernel_size = blk.active_kernel_size
            y, coef, Ho, Wo = self.dw(src, blk.depth_conv, C, blk.active_kernel_size, B, Hi, Wi)
            src = (y, coef, True)
        pl = blk.point_linear
        pl.conv.active_out_channel = blk.active_out_channel      # layers.py:453 ...
        pl.conv.active_out_channel = pl.active_out_channel       # ... overwritten by the layer's own width (layers.py:353)
        N = pl.conv.active_out_channel
        y, coef = self.pw(src, pl, C, N, B * Ho * Wo)
        res = x if (blk.stride == 1 and Cin == blk.active_out_channel) else None
        if res is not None and (N != Cin or (Ho, Wo) != (Hi, Wi)):
            raise RuntimeError(f"residual add of a {Cin}-channel input to a {N}-channel output")
        return self.apply(y.view(B, Ho, Wo, N), coef, False, res=res, skip=skip,
                          out_hw=self._decoder_hw(Ho, Wo, skip))

    @staticmethod
    def _decoder_hw(Ho, Wo, skip):
        # `x + skip[..., -h:, -w:]` keeps the decoder tensor's size; a smaller skip is the reference's broadcast error
        if skip is not None and (skip.shape[1] < Ho or skip.shape[2] < Wo):
            raise RuntimeError(f"The size of the decoder tensor ({Ho} x {Wo}) must match the size of the encoder skip "
                               f"({skip.shape[1]} x {skip.shape[2]})")
        return Ho, Wo

    def output(self):
        net, B, H, W = self.net, self.B, self.H, self.W
        if net.runtime_depth[0] < 1:
            raise NotImplementedError("runtime_depth[0] = 0 drops the first and last layers; the blocks behind them cannot "
                                      "take a one-channel tensor in the reference either")
        first = net.encoder[0]
        first.conv.active_out_channel = first.active_out_channel
        C = first.active_out_channel
        w = first.conv.conv.weight
        if C > w.shape[0]:
            raise ValueError(f"{C} active channels exceed the first layer's {w.shape[0]}")
        hp, wp = net.H_padding, net.W_padding
        Ho, Wo = (H + hp - 1) // 2 + 1, (W + wp - 1) // 2 + 1
        y = self.new(B, Ho, Wo, C)
        _lib.call("noise_ofa_first_conv", ptr(self.x), ptr(w), ptr(y), ptr(self.slab), ctypes.byref(self.rows), B, H, W, hp, wp,
                  C, _lib.stream(), nbytes=4 * (B * H * W + y.numel()))
        x = self.apply(y, self.bn(first.bn, C, B * Ho * Wo), True)
        # encoder (layers.py:648-655)
        downs = []
        for stage, idxs in enumerate(net.block_group_info):
            for idx in idxs[:net.runtime_depth[stage]]:
                if idx > 0:
                    x = self.block(net.encoder[idx], x)
            downs.append(x)
        # decoder (layers.py:657-664): add the stage's skip, run the mirrors of the stage's LAST `depth` blocks
        nblocks = len(net.decoder) - 1
        x = self.apply(x, None, False, skip=downs[-1], out_hw=self._decoder_hw(x.shape[1], x.shape[2], downs[-1]))
        for stage in range(len(net.block_group_info) - 1, 0, -1):
            active = net.block_group_info[stage][::-1][:net.runtime_depth[stage]]
            nxt = downs[stage - 1]
            for j, idx in enumerate(active):
                x = self.block(net.decoder[nblocks - idx], x, skip=nxt if j == len(active) - 1 else None)
            if not active:
                x = self.apply(x, None, False, skip=nxt, out_hw=self._decoder_hw(x.shape[1], x.shape[2], nxt))
        last = net.decoder[nblocks]
        last.conv.active_out_channel = last.active_out_channel
        Bx, Hi, Wi, C = x.shape
        w = last.conv.conv.weight
        if last.active_out_channel != 1 or C > w.shape[0]:
            raise ValueError(f"last layer: {C} -> {last.active_out_channel} channels against weights {tuple(w.shape)}")
        Hy, Wy = 2 * Hi - 1, 2 * Wi - 1
        y = self.new(B, Hy, Wy)
        _lib.call("noise_ofa_last_convT", ptr(x), None, 0, ptr(w), ptr(y), ptr(self.slab), ctypes.byref(self.rows), B, Hi, Wi, C,
                  _lib.stream(), nbytes=4 * (x.numel() + y.numel()))
        coef = self.bn(last.bn, 1, B * Hy * Wy)
        Hout, Wout = min(H, Hy), min(W, Wy)       # `x[:, :, -H:, -W:]` of a smaller tensor is the tensor itself
        out = self.new(B, 1, Hout, Wout)
        _lib.call("noise_ofa_final", ptr(y), ptr(coef), 1, ptr(out), B, Hy, Wy, Hout, Wout, _lib.stream(),
                  nbytes=8 * out.numel())
        return out
