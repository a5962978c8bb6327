#!/usr/bin/env python3
"""Golden fixture G8: the reference's `OFASpectralNets` (train/layers.py) run on CPU.

Runs ONLY in the build container (it needs /root/reference; layers.py imports with torch and numpy alone, no stand-ins).
Nothing but inputs, outputs and per-tensor checksums is stored: the tests rebuild the 15.5 M parameters from the seed
(`torch.manual_seed(seed)` + constructor, then `perturb()` below, which both sides run) and compare.

Usage:  python tests/golden/make_golden_ofa.py
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/train/layers.py"
OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (constructor kwargs, seed, input (B, H, W), sub-network choice)
    "default_k7": ({}, 11, (2, 97, 65), {}),
    "default_k3": ({}, 12, (2, 80, 49), {"ks": 3}),
    "default_k5_narrow": ({}, 13, (3, 65, 33), {"ks": 5, "wid": 0}),
    "small_k35": ({"base_stage_width": [16, 8, 16, 16, 24, 24, 32, 32, 64], "kernel_size_list": [3, 5],
                   "expand_ratio_list": [2, 4], "width_mult_list": [0.5, 1.0]}, 14, (2, 70, 41), {"ks": 3, "wid": 1}),
}


def perturb(net, seed):
    """Move every parameter the constructor leaves at a trivial value (transform matrices = identity, BatchNorm weight = 1,
    bias = 0, running statistics = 0 / 1) so that each of them matters; both the generator and the tests call this."""
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("_matrix"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("bn.weight"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif name.endswith("bn.bias"):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for name, b in net.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))


def choose(net, ks=None, wid=None):
    """Sub-network choice through the attributes the reference's forward reads (its own `set_active_subnet` fails on an
    undefined attribute): kernel size per block; width index on every module that owns a width list of that length."""
    for m in net.modules():
        if ks is not None and hasattr(m, "active_kernel_size") and hasattr(m, "point_linear"):
            m.active_kernel_size = ks
        if wid is not None and hasattr(m, "out_channel_list") and len(m.out_channel_list) > 1:
            m.active_out_channel = m.out_channel_list[wid]
            if hasattr(m, "point_linear"):
                m.point_linear.active_out_channel = m.out_channel_list[wid]


def main():
    spec = importlib.util.spec_from_file_location("ref_layers", REF)
    L = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(L)
    out = {}
    keys = None
    for name, (kw, seed, (B, H, W), sub) in CASES.items():
        torch.manual_seed(seed)
        net = L.OFASpectralNets(**kw)
        if name == "default_k7":
            sd = net.state_dict()
            keys = [f"{k} {tuple(v.shape)}" for k, v in sd.items()]
            out["default_checksums"] = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
        perturb(net, seed)
        choose(net, **sub)
        x = torch.randn(B, 1, H, W, generator=torch.Generator().manual_seed(seed + 2000)) * 1.5 - 0.5
        net.train()
        y_train = net(x)                       # first call: padding search (whole forwards) + the forward itself
        y_train2 = net(x)                      # second call: statistics moved once more
        first_bn, deep_bn = net.encoder[0].bn.bn, net.encoder[-1].point_linear.bn.bn
        out[f"{name}_x"] = x.numpy()
        out[f"{name}_train"] = y_train.detach().numpy()
        out[f"{name}_train2"] = y_train2.detach().numpy()
        out[f"{name}_padding"] = np.array([net.H_padding, net.W_padding])
        out[f"{name}_nbt"] = np.array([int(first_bn.num_batches_tracked), int(deep_bn.num_batches_tracked)])
        out[f"{name}_first_running"] = np.stack([first_bn.running_mean.numpy(), first_bn.running_var.numpy()])
        out[f"{name}_deep_running"] = np.stack([deep_bn.running_mean.numpy(), deep_bn.running_var.numpy()])
        net.eval()
        out[f"{name}_eval"] = net(x).detach().numpy()
        print(name, tuple(y_train.shape), "padding", net.H_padding, net.W_padding, "nbt", out[f"{name}_nbt"],
              "max", float(y_train.abs().max()), float(out[f"{name}_eval"].max()))
    np.savez_compressed(os.path.join(OUT, "g8_ofa.npz"), **out)
    with open(os.path.join(OUT, "g8_ofa_keys.txt"), "w") as f:
        f.write("\n".join(keys) + "\n")


if __name__ == "__main__":
    main()
