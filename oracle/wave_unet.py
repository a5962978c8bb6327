"""CPU oracle (test infrastructure, never the product path) for the forward pass of the waveform-domain U-Net.

Restates `Base.forward` / `_forward` / `_normalize` / `_search_padding` (train/models.py:576-639), `InvertedResidual`
(:348-407), the transformer encoder (:14-225) and the layer stack `VADMobileNetV2.__init__` builds (:1096-1218) with
torch's own operators on the CPU, in the precision of the tensors it is given.  It walks any module tree with the
reference's child layout (`encoder[i]` / `decoder[i]` = [conv, BatchNorm, ReLU6] or a block with `.conv` =
[expand?, depthwise, 1x1 conv, BatchNorm] and `.use_res_connect`; `bottleneck_attention.layer_stack[i].slf_attn.{w_qs,
w_ks, w_vs, fc, layer_norm}`, `.pos_ffn.{w_1, w_2, layer_norm}`, `.layer_norm`; `vad`) and changes the same state the
reference changes (running statistics, batch counters, the padding found by the search).  Pinned by
tests/golden/g9_wave.npz, written by the reference itself (tests/golden/make_golden_wave.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def batch_norm(x, bn, training):
    """nn.BatchNorm2d.forward: batch statistics + running-statistics update in train mode."""
    factor = 0.0
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
        factor = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
    if x.dtype != bn.running_mean.dtype:          # float64 yardstick: statistics are updated through copies
        rm, rv = bn.running_mean.to(x.dtype), bn.running_var.to(x.dtype)
        y = F.batch_norm(x, rm, rv, bn.weight.to(x.dtype), bn.bias.to(x.dtype), training, factor, bn.eps)
        bn.running_mean.copy_(rm)
        bn.running_var.copy_(rv)
        return y
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, factor, bn.eps)


def conv_norm_act(seq, x, training):
    """ConvNormActivation (torchvision) / ConvTranspose2dNormActivation (models.py:316-345): conv [-> BN] [-> ReLU6]."""
    conv = seq[0]
    w = conv.weight.to(x.dtype)
    b = conv.bias.to(x.dtype) if conv.bias is not None else None
    if isinstance(conv, nn.ConvTranspose2d):
        x = F.conv_transpose2d(x, w, b, conv.stride, conv.padding, 0, conv.groups)
    else:
        x = F.conv2d(x, w, b, conv.stride, conv.padding, 1, conv.groups)
    for mod in list(seq)[1:]:
        if isinstance(mod, nn.BatchNorm2d):
            x = batch_norm(x, mod, training)
        elif isinstance(mod, nn.ReLU6):
            x = F.relu6(x)
    return x


def block(blk, x, training):
    """InvertedResidual.forward (models.py:403-407)."""
    y = x
    for mod in blk.conv:
        if isinstance(mod, nn.Sequential):
            y = conv_norm_act(mod, y, training)
        elif isinstance(mod, nn.BatchNorm2d):
            y = batch_norm(y, mod, training)
        else:
            y = F.conv2d(y, mod.weight.to(y.dtype))
    return x + y if blk.use_res_connect else y


def layer_norm(x, ln):
    return F.layer_norm(x, (x.shape[-1],), ln.weight.to(x.dtype), ln.bias.to(x.dtype), ln.eps)


def transformer(enc, x, keep_masks=None, p=0.0):
    """TransformerEncoder.forward with n_position = 0 (models.py:204-225) around MultiHeadAttention (:60-92),
    ScaledDotProductAttention (:26-36) and PositionwiseFeedForward (:105-118).  `keep_masks[i]` (B, heads, n, n) is the
    attention dropout of layer i (kept weights scaled by 1 / (1 - p))."""
    x = layer_norm(x, enc.layer_norm)
    B, n, _ = x.shape
    for i, layer in enumerate(enc.layer_stack):
        sa, ff = layer.slf_attn, layer.pos_ffn
        H, dk = sa.n_head, sa.d_k
        q = F.linear(x, sa.w_qs.weight.to(x.dtype)).view(B, n, H, dk).transpose(1, 2)
        k = F.linear(x, sa.w_ks.weight.to(x.dtype)).view(B, n, H, dk).transpose(1, 2)
        v = F.linear(x, sa.w_vs.weight.to(x.dtype)).view(B, n, H, dk).transpose(1, 2)
        attn = torch.softmax(torch.matmul(q / dk ** 0.5, k.transpose(2, 3)), dim=-1)
        if keep_masks is not None:
            attn = attn * keep_masks[i].to(x.dtype) / (1.0 - p)
        o = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, n, -1)
        x = layer_norm(F.linear(o, sa.fc.weight.to(x.dtype)) + x, sa.layer_norm)
        h = F.relu(F.linear(x, ff.w_1.weight.to(x.dtype), ff.w_1.bias.to(x.dtype)))
        x = layer_norm(F.linear(h, ff.w_2.weight.to(x.dtype), ff.w_2.bias.to(x.dtype)) + x, ff.layer_norm)
    return x


def normalize(x):
    """Base._normalize (models.py:614-622)."""
    mx = x.abs().reshape(x.size(0), x.size(1), -1).max(dim=-1, keepdim=True)[0].unsqueeze(2) + 1e-6
    return x / mx, mx


def _pass(m, x, training, keep_masks=None):
    """Base._forward after the search (models.py:588-612)."""
    H, W = x.shape[-2:]
    x = F.pad(x, (m.W_padding, 0, m.H_padding, 0), value=m.pad_value)
    downs = []
    for i, mod in enumerate(m.encoder):
        x = conv_norm_act(mod, x, training) if i == 0 else block(mod, x, training)
        downs.append(x)
    downs = downs[::-1]
    b, c, h, w = x.shape
    t = x.reshape(b, c, -1).permute(0, 2, 1)
    t = transformer(m.bottleneck_attention, t, keep_masks if training else None, m.attn_dropout)
    vad = F.linear(t[:, 0, :], m.vad.weight.to(t.dtype), m.vad.bias.to(t.dtype))
    x = t.permute(0, 2, 1).reshape(b, c, h, w)
    last = len(m.decoder) - 1
    for i, mod in enumerate(m.decoder):
        skip = downs[i]
        x = x + skip[:, :, -x.shape[-2]:, -x.shape[-1]:]
        x = conv_norm_act(mod, x, training) if i == last else block(mod, x, training)
    return x[:, :, -H:, -W:], vad


def full_forward(m, x, training, keep_masks=None):
    """Base.forward (models.py:576-581)."""
    xn, mx = normalize(x)
    y, vad = forward(m, xn, training, keep_masks)
    return y * torch.sigmoid(1e4 * vad[..., None, None]) * mx


@torch.no_grad()
def forward(m, xn, training, keep_masks=None):
    """Base._forward including the one-off padding search (models.py:583-612, :628-639), whose whole forwards move the
    running statistics in train mode.  Attention dropout needs explicit masks (none = no dropout)."""
    if not m.searching and not m.found:
        m.searching = True
        H, W = xn.shape[-2:]
        out = full_forward(m, xn, training)
        while out.shape[-2] < H:
            m.H_padding += 1
            out = full_forward(m, xn, training)
        while out.shape[-1] < W:
            m.W_padding += 1
            out = full_forward(m, xn, training)
        m.found = True
    return _pass(m, xn, training, keep_masks)


@torch.no_grad()
def shared_step(m, noisy, clean, mr_stft_lambda, fft_sizes, hop_lengths, win_lengths, segment=8000):
    """Base._shared_step (models.py:641-665) in eval mode: (total, {waveform, spectral, vad}) as Python floats."""
    from . import mrstft
    noisy = noisy.reshape(-1, 1, 1, segment)
    clean = clean.reshape(-1, 1, 1, segment)
    xn, mx = normalize(noisy)
    filt, vad = forward(m, xn, False)
    enhanced = filt * mx
    total, parts, _, _ = mrstft.waveform_loss(enhanced.reshape(-1, segment).double().numpy(),
                                              clean.reshape(-1, segment).double().numpy(), vad.double().numpy(), mr_stft_lambda,
                                              fft_sizes, hop_lengths, win_lengths)
    return total, parts


def train_step(m, noisy, clean, keep_masks, mr_stft_lambda, fft_sizes, hop_lengths, win_lengths, dtype=torch.float32, segment=8000):
    """Base.training_step (models.py:641-682) in train mode with gradients: ([total, waveform, spectral, vad], {name: grad}).
    The network runs under autograd in `dtype`; the loss and its gradient with respect to the enhanced waveform and the VAD
    logits come from the numpy restatement (oracle/mrstft.py) and are chained into the graph."""
    from . import mrstft
    noisy = noisy.reshape(-1, 1, 1, segment).to(dtype)
    clean = clean.reshape(-1, 1, 1, segment)
    xn, mx = normalize(noisy)
    assert m.found, "run the padding search first"
    with torch.enable_grad():
        filt, vad = _pass(m, xn, True, keep_masks)
        enhanced = filt * mx
    total, parts, d_enh, d_vad = mrstft.waveform_loss(enhanced.detach().reshape(-1, segment).double().numpy(),
                                                      clean.reshape(-1, segment).double().numpy(),
                                                      vad.detach().double().numpy(), mr_stft_lambda, fft_sizes, hop_lengths,
                                                      win_lengths)
    for p in m.parameters():
        p.grad = None
    torch.autograd.backward([enhanced, vad], [torch.from_numpy(d_enh).reshape(enhanced.shape).to(dtype),
                                              torch.from_numpy(d_vad).reshape(vad.shape).to(dtype)])
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    return [total, parts["l1"], parts["mrstft"], parts["vad"]], grads
