"""CPU oracle (test infrastructure, never the product path) for the once-for-all spectral U-Net forward.

Restates train/layers.py:108-460 (dynamic depthwise / 1x1 / BatchNorm / inverted-residual block) and :627-668
(`OFASpectralNets._search_padding`, `.forward`) with torch's own operators on the CPU, in the precision of the tensors it
is given (fp32 like the reference, or fp64 as a yardstick).  It works on any module tree that has the reference's
attribute names (`encoder`, `decoder`, `block_group_info`, `runtime_depth`, `H_padding`, ...; each block's
`inverted_bottleneck` / `depth_conv` / `point_linear` with `.conv.conv`, `.bn.bn`, the `<a>to<b>_matrix` parameters and
the `active_*` attributes) and changes the same state the reference changes (running statistics, batch counters, the
padding found by the search).  Pinned by tests/golden/g8_ofa.npz, written by the reference itself
(tests/golden/make_golden_ofa.py)."""
import torch
import torch.nn.functional as F


def make_divisible(v, divisor, min_val=None):
    """layers.py:57-74."""
    if min_val is None:
        min_val = divisor
    new_v = max(min_val, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def batch_norm(x, dbn):
    """DynamicBatchNorm2d.forward (layers.py:284-313): the leading x.size(1) channels of the parameters and buffers."""
    bn, c = dbn.bn, x.size(1)
    factor = 0.0
    if bn.training and bn.track_running_stats:
        bn.num_batches_tracked += 1
        factor = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
    rm, rv = bn.running_mean[:c], bn.running_var[:c]
    if x.dtype != rm.dtype:                       # fp64 yardstick: statistics are updated through copies
        rm2, rv2 = rm.to(x.dtype), rv.to(x.dtype)
        y = F.batch_norm(x, rm2, rv2, bn.weight[:c].to(x.dtype), bn.bias[:c].to(x.dtype),
                         bn.training or not bn.track_running_stats, factor, bn.eps)
        rm.copy_(rm2)
        rv.copy_(rv2)
        return y
    return F.batch_norm(x, rm, rv, bn.weight[:c], bn.bias[:c], bn.training or not bn.track_running_stats, factor, bn.eps)


def active_filter(sep, channels, kernel_size):
    """DynamicSeparableConv2d.get_active_filter (layers.py:162-196)."""
    w = sep.conv.weight[:channels]
    ks = sorted(set(sep.kernel_size_list))
    kmax = max(ks)
    cur, size = w, kmax
    for i in range(len(ks) - 1, 0, -1):
        if ks[i] <= kernel_size:
            break
        target = ks[i - 1]
        s = size // 2 - target // 2
        crop = cur[:, :, s:s + target, s:s + target].reshape(-1, target * target)
        cur = F.linear(crop, getattr(sep, f"{ks[i]}to{target}_matrix")).view(w.size(0), w.size(1), target, target)
        size = target
    if size != kernel_size:          # no transform reached this size: plain centre crop
        s = size // 2 - kernel_size // 2
        cur = cur[:, :, s:s + kernel_size, s:s + kernel_size]
    return cur.contiguous()


def conv_norm_act(layer, x, relu6):
    """DynamicConv2dNormActivation.forward (layers.py:352-360) around DynamicPointConv2d.forward (:253-272)."""
    layer.conv.active_out_channel = layer.active_out_channel
    n, c = layer.conv.active_out_channel, x.size(1)
    conv = layer.conv.conv
    pad = (layer.kernel_size - 1) // 2
    if isinstance(conv, torch.nn.ConvTranspose2d):
        y = F.conv_transpose2d(x, conv.weight[:c, :n].to(x.dtype).contiguous(), None, conv.stride, pad)
    else:
        y = F.conv2d(x, conv.weight[:n, :c].to(x.dtype).contiguous(), None, conv.stride, pad)
    if getattr(layer, "use_bn", True):
        y = batch_norm(y, layer.bn)
    return F.relu6(y) if relu6 else y


def block(blk, x):
    """DynamicInvertedResidual.forward (layers.py:436-460)."""
    x0, cin = x, x.size(1)
    if blk.inverted_bottleneck is not None:
        blk.inverted_bottleneck.conv.active_out_channel = make_divisible(round(cin * blk.active_expand_ratio), 8)
        x = conv_norm_act(blk.inverted_bottleneck, x, True)
    if blk.depth_conv is not None:
        sep = blk.depth_conv.conv
        sep.active_kernel_size = k = blk.active_kernel_size
        f = active_filter(sep, x.size(1), k).to(x.dtype)
        if isinstance(sep.conv, torch.nn.ConvTranspose2d):
            x = F.conv_transpose2d(x, f, None, sep.conv.stride, k // 2, 0, x.size(1))
        else:
            x = F.conv2d(x, f, None, sep.conv.stride, k // 2, 1, x.size(1))
        x = F.relu6(batch_norm(x, blk.depth_conv.bn))
    blk.point_linear.conv.active_out_channel = blk.active_out_channel
    x = conv_norm_act(blk.point_linear, x, False)
    if blk.stride == 1 and cin == blk.active_out_channel:
        x = x + x0
    return x


def _pass(net, x):
    """The body of OFASpectralNets.forward after the search (layers.py:645-668)."""
    H, W = x.shape[-2:]
    x = F.pad(x, (net.W_padding, 0, net.H_padding, 0), value=0.0)
    downs = []
    for stage, idxs in enumerate(net.block_group_info):
        for idx in idxs[:net.runtime_depth[stage]]:
            x = conv_norm_act(net.encoder[0], x, True) if idx == 0 else block(net.encoder[idx], x)
        downs.append(x)
    last = len(net.decoder) - 1
    for stage in range(len(net.block_group_info) - 1, -1, -1):
        skip = downs[stage]
        x = x + skip[:, :, -x.shape[-2]:, -x.shape[-1]:]
        for idx in net.block_group_info[stage][::-1][:net.runtime_depth[stage]]:
            x = conv_norm_act(net.decoder[last], x, True) if idx == 0 else block(net.decoder[last - idx], x)
    return x[:, :, -H:, -W:]


@torch.no_grad()
def forward(net, x):
    """OFASpectralNets.forward (layers.py:640-668) including the one-off padding search (:627-638), whose extra forwards
    move the running statistics in train mode."""
    if not net.searching and not net.found:
        net.searching = True
        H, W = x.shape[-2:]
        out = _pass(net, x)
        while out.shape[-2] < H:
            net.H_padding += 1
            out = _pass(net, x)
        while out.shape[-1] < W:
            net.W_padding += 1
            out = _pass(net, x)
        net.found = True
    return _pass(net, x)
