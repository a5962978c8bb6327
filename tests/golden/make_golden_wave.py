#!/usr/bin/env python3
"""Golden fixture G9: the reference's waveform-domain `VADMobileNetV2` (train/models.py:1096-1218 under `Base`, :541-612)
run on CPU in the configuration of train/configs/mobilenetv2.yaml (1x3 kernels, stride (1,2), repeated residual blocks,
two encoder layers, d_model 128 / 6 heads / d_k 21) and in a narrow variant of it.

Runs ONLY in the build container (it needs /root/reference); the reference is imported untouched behind the stand-ins of
make_golden.py.  Nothing but inputs and outputs is stored: the tests rebuild the parameters from the seed
(`torch.manual_seed(seed)` + constructor, then `perturb()` below, which both sides run).

Usage:  python tests/golden/make_golden_wave.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import make_golden as G  # noqa: E402  (stand-ins + reference path)

YAML = dict(nhead=6, num_layers=2, dropout=0.0, pad_value=0.0, mr_stft_lambda=0.1,
            layer_config=[[1, 32, 1, [1, 3], [1, 2]], [1, 16, 1, [1, 3], [1, 1]], [6, 24, 2, [1, 3], [1, 2]],
                          [6, 32, 2, [1, 3], [1, 2]], [6, 64, 2, [1, 3], [1, 1]], [6, 96, 2, [1, 3], [1, 2]],
                          [6, 128, 1, [1, 3], [1, 2]]])
NARROW = dict(nhead=4, num_layers=2, dropout=0.0, pad_value=0.0,
              layer_config=[[1, 16, 1, [1, 3], [1, 2]], [1, 8, 1, [1, 3], [1, 1]], [4, 16, 2, [1, 3], [1, 2]],
                            [4, 24, 2, [1, 3], [1, 1]], [4, 32, 1, [1, 3], [1, 2]]])
SQUARE = dict(nhead=4, num_layers=1, dropout=0.0, pad_value=0.0,      # 3x3 kernels on a real image, residual blocks
              layer_config=[[1, 16, 1, [3, 3], [2, 2]], [1, 8, 1, [3, 3], [1, 1]], [4, 16, 2, [3, 3], [2, 2]],
                            [4, 32, 2, [3, 3], [2, 2]]])

CASES = {
    # name: (constructor kwargs, seed, first input shape, second input shape (same model, padding already found))
    "yaml": (YAML, 31, (3, 1, 1, 8000), (2, 1, 1, 5003)),
    "narrow": (NARROW, 32, (4, 1, 1, 4000), (2, 1, 1, 1777)),
    "square": (SQUARE, 33, (2, 1, 40, 52), (1, 1, 33, 47)),
}


def kwargs(cfg):
    return {k: ([list(map(lambda e: list(e) if isinstance(e, list) else e, r)) for r in v] if k == "layer_config" else v)
            for k, v in cfg.items()}


def perturb(m, seed, linear_gain=8.0):
    """Move the tensors the constructor leaves trivial (BatchNorm / LayerNorm affine, running statistics, biases) and tame
    the last layer so the outputs stay of order one; both the generator and the tests call this.  The training fixture
    keeps the constructor's scale of the Linear layers (`linear_gain=1`): with the larger weights the attention saturates
    and the decoder's gradients are 20-30 % rounding noise in fp32 (float64 evaluation against the reference's own fp32
    gradients), which pins nothing."""
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.weight.copy_(1 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
                mod.running_mean.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
                mod.running_var.copy_(1 + 0.3 * torch.rand(mod.bias.shape, generator=g))
            elif isinstance(mod, nn.LayerNorm):
                mod.weight.copy_(1 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
            elif isinstance(mod, nn.Linear):
                mod.weight.mul_(linear_gain)     # the constructor's std of 0.01 leaves attention and FFN negligible
                if mod.bias is not None:
                    mod.bias.copy_(0.05 * torch.randn(mod.bias.shape, generator=g))
        m.vad.weight.copy_(0.02 * torch.randn(m.vad.weight.shape, generator=g))    # a gate that is neither 0 nor 1 ...
        m.vad.bias.fill_(1e-5)                                                     # ... for some clips (sigmoid(1e4 * v))
        m.decoder[-1][0].weight.mul_(0.3)
        m.decoder[-1][0].bias.fill_(0.01)


def wave(seed, shape):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * 0.1
    x[0] *= 3.0                                   # per-clip maxima that differ
    return x.clamp(-1, 1)


def train_fixture(models, g9):
    """G10: one training step (Base.training_step, models.py:641-682) per case: losses, attention keep masks, the gradient
    of every parameter (whole tensors for the narrow case, per-tensor sums + norms and a few whole tensors for the wide
    one)."""
    out = {}
    for name in ("narrow", "yaml"):
        cfg, seed, s1, _ = CASES[name]
        torch.manual_seed(seed)
        m = models.VADMobileNetV2(**kwargs(cfg))
        perturb(m, seed, linear_gain=1.0)
        with torch.no_grad():
            m.vad.bias.copy_(torch.from_numpy(g9[f"{name}_vad_bias"]))
        x1 = wave(seed + 2000, s1)
        clean = torch.from_numpy(g9[f"{name}_val_clean"])
        m.eval()
        with torch.no_grad():
            m(x1.clone())                         # padding search in eval mode: the running statistics stay put
        m.train()
        masks = []
        for lay in m.bottleneck_attention.layer_stack:
            lay.slf_attn.attention.dropout.register_forward_hook(lambda mod, inp, o: masks.append((o != 0).numpy().copy()))
        torch.manual_seed(seed + 9)
        loss = m.training_step((x1.clone(), clean.clone(), None), 0)
        loss.backward()
        out[f"{name}_losses"] = np.array([float(loss)] + [m._logged[f"train_{k}"] for k in
                                                          ("waveform_loss", "spectral_loss", "vad_loss")])
        for i, k in enumerate(masks):
            out[f"{name}_keep{i}"] = np.packbits(k.reshape(-1))
            out[f"{name}_keep{i}_shape"] = np.array(k.shape)
        names, stats = [], []
        whole = ("encoder.0.0.weight", "encoder.0.1.weight", "encoder.2.conv.0.0.weight", "encoder.2.conv.1.0.weight",
                 "encoder.3.conv.2.weight", "decoder.0.conv.1.0.weight", "bottleneck_attention.layer_stack.0.slf_attn.w_qs.weight",
                 "bottleneck_attention.layer_stack.1.pos_ffn.w_1.bias", "bottleneck_attention.layer_norm.weight", "vad.weight",
                 "vad.bias", f"decoder.{len(m.decoder) - 1}.0.weight", f"decoder.{len(m.decoder) - 1}.0.bias")
        for k, p in m.named_parameters():
            gk = p.grad.double()
            names.append(k)
            stats.append([float(gk.sum()), float(gk.abs().sum()), float(gk.norm())])
            if name == "narrow" or k in whole:
                out[f"{name}_grad/{k}"] = p.grad.numpy().copy()
        out[f"{name}_grad_names"] = np.array(names)
        out[f"{name}_grad_stats"] = np.array(stats)
        for key in ("encoder.0.1", "decoder.0.conv.3"):
            out[f"{name}_buf/{key}.running_var"] = dict(m.named_buffers())[key + ".running_var"].numpy().copy()
        print(name, "train losses", out[f"{name}_losses"], "grad tensors", len(names))
    np.savez_compressed(os.path.join(OUT, "g10_wave_train.npz"), **out)
    print("wrote g10_wave_train.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


def main():
    G._install_standins()
    sys.path.insert(0, os.path.join(G.REF, "train"))
    import models  # noqa: the reference, untouched

    out = {}
    for name, (cfg, seed, s1, s2) in CASES.items():
        torch.manual_seed(seed)
        m = models.VADMobileNetV2(**kwargs(cfg))
        if name == "yaml":
            sd = m.state_dict()
            with open(os.path.join(OUT, "g9_wave_keys.txt"), "w") as f:
                f.write("\n".join(f"{k} {tuple(v.shape)}" for k, v in sd.items()) + "\n")
            out["yaml_checksums"] = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
        perturb(m, seed)
        x1, x2 = wave(seed + 2000, s1), wave(seed + 3000, s2)
        m.eval()
        with torch.no_grad():
            # the gate is sigmoid(1e4 * vad): centre the logits of this input on zero so that it is neither 0 nor 1
            # (the bias is stored: it is an input of the test like the seed)
            _, v0 = m._forward(m._normalize(x1.clone()))
            m.vad.bias.sub_(v0.mean())
            out[f"{name}_vad_bias"] = m.vad.bias.numpy().copy()
            y1 = m(x1.clone())                    # padding found above (models.py:585-586 runs the search once)
            pad = (m.H_padding, m.W_padding)
            y2 = m(x2.clone())                    # another length with the padding (and cropped skips) of the first
            xn = m._normalize(x1.clone())
            filt, vad = m._forward(xn)
        out.update({f"{name}_x1": x1.numpy(), f"{name}_x2": x2.numpy(), f"{name}_eval_y1": y1.numpy(),
                    f"{name}_eval_y2": y2.numpy(), f"{name}_eval_filters": filt.numpy(), f"{name}_eval_vad": vad.numpy(),
                    f"{name}_padding": np.array(pad)})
        # Base._shared_step through validation_step (models.py:641-685): the batch as 8000-sample segments, losses logged
        if x1.numel() % 8000 == 0:
            clean = wave(seed + 4000, s1) * 0.5
            clean.view(-1, 8000)[0] = 0.0                    # a silent segment: VAD label 0
            with torch.no_grad():
                loss = m.validation_step((x1.clone(), clean.clone(), None), 0)
            out[f"{name}_val_clean"] = clean.numpy()
            out[f"{name}_val_losses"] = np.array([float(loss)] + [m._logged[f"val_{k}"] for k in
                                                                  ("loss", "waveform_loss", "spectral_loss", "vad_loss")])
            print(name, "val losses", out[f"{name}_val_losses"])
        # train mode on the same model (padding known): batch statistics, attention dropout (keep masks recorded per
        # encoder layer), running statistics moved once
        masks = []
        hooks = [lay.slf_attn.attention.dropout.register_forward_hook(
            lambda mod, inp, o: masks.append((o != 0).numpy().copy())) for lay in m.bottleneck_attention.layer_stack]
        m.train()
        torch.manual_seed(seed + 7)
        with torch.no_grad():
            ft, vt = m._forward(m._normalize(x1.clone()))
        for h in hooks:
            h.remove()
        assert len(masks) == len(m.bottleneck_attention.layer_stack)
        out[f"{name}_train_filters"] = ft.numpy()
        out[f"{name}_train_vad"] = vt.numpy()
        for i, k in enumerate(masks):
            out[f"{name}_keep{i}"] = np.packbits(k.reshape(-1))
            out[f"{name}_keep{i}_shape"] = np.array(k.shape)
        bufs = dict(m.named_buffers())
        for key in ("encoder.0.1", "encoder.2.conv.1.1", "decoder.0.conv.3", f"decoder.{len(m.decoder) - 2}.conv.1.1"):
            if key + ".running_mean" in bufs:
                out[f"{name}_buf/{key}.running_mean"] = bufs[key + ".running_mean"].numpy().copy()
                out[f"{name}_buf/{key}.running_var"] = bufs[key + ".running_var"].numpy().copy()
                out[f"{name}_buf/{key}.num_batches_tracked"] = bufs[key + ".num_batches_tracked"].numpy().copy()
        # a fresh model whose FIRST call is in train mode: the padding search runs whole forwards (models.py:628-639),
        # each of which moves the running statistics; attention dropout switched off so that the result is reproducible
        torch.manual_seed(seed)
        f = models.VADMobileNetV2(**kwargs(cfg))
        perturb(f, seed)
        with torch.no_grad():
            f.vad.bias.copy_(m.vad.bias)
        for lay in f.bottleneck_attention.layer_stack:
            lay.slf_attn.attention.dropout.p = 0.0
        f.train()
        with torch.no_grad():
            ff, vf = f._forward(f._normalize(x1.clone()))
        out[f"{name}_fresh_train_filters"] = ff.numpy()
        out[f"{name}_fresh_train_vad"] = vf.numpy()
        fb = dict(f.named_buffers())
        out[f"{name}_fresh/encoder.0.1.running_mean"] = fb["encoder.0.1.running_mean"].numpy().copy()
        out[f"{name}_fresh/encoder.0.1.running_var"] = fb["encoder.0.1.running_var"].numpy().copy()
        out[f"{name}_fresh/encoder.0.1.num_batches_tracked"] = fb["encoder.0.1.num_batches_tracked"].numpy().copy()
        out[f"{name}_fresh/decoder.0.conv.3.running_mean"] = fb["decoder.0.conv.3.running_mean"].numpy().copy()
        print(name, "padding", pad, "eval |y|", float(y1.abs().mean()), "train |f|", float(ft.abs().mean()), "eval |f|", float(filt.abs().mean()),
              "vad", vad.flatten().tolist(), "searches", int(fb["encoder.0.1.num_batches_tracked"]))
    train_fixture(models, out)
    np.savez_compressed(os.path.join(OUT, "g9_wave.npz"), **out)
    print("wrote g9_wave.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
