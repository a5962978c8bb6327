"""The once-for-all spectral U-Net forward (train/layers.py:496-668) on the HIP kernels against the reference's own outputs
(tests/golden/g8_ofa.npz, written by make_golden_ofa.py from the reference run on CPU): train mode (batch statistics, the
padding search's extra forwards, running-statistics updates) and eval mode, kernel sizes 7 / 5 / 3 through the transform
matrices, full and reduced widths, a small custom width table."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

# The network is ~150 BatchNorm layers deep in fp32 and train-mode statistics at the bottom of the U come from a few dozen
# pixels, so rounding differences grow on the way: outputs agree to TOL of the largest reference value.
TOL = 2e-3


def _build(name):
    from make_golden_ofa import CASES, choose, perturb
    from noise_amd.ofa import OFASpectralNets
    kw, seed, _, sub = CASES[name]
    torch.manual_seed(seed)
    net = OFASpectralNets(**kw)
    perturb(net, seed)
    choose(net, **sub)
    return net.to("cuda")


@pytest.mark.parametrize("name", ["default_k7", "default_k3", "default_k5_narrow", "small_k35"])
def test_forward_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "g8_ofa.npz"))
    net = _build(name)
    x = torch.from_numpy(g[f"{name}_x"]).cuda()
    net.train()
    y1 = net(x)
    assert [net.H_padding, net.W_padding] == list(g[f"{name}_padding"])
    y2 = net(x)
    first_bn, deep_bn = net.encoder[0].bn.bn, net.encoder[-1].point_linear.bn.bn
    assert [int(first_bn.num_batches_tracked), int(deep_bn.num_batches_tracked)] == list(g[f"{name}_nbt"])
    for got, key in ((y1, "train"), (y2, "train2")):
        want = g[f"{name}_{key}"]
        assert tuple(got.shape) == want.shape
        assert np.abs(got.cpu().numpy() - want).max() <= TOL * np.abs(want).max(), key
    np.testing.assert_allclose(torch.stack([first_bn.running_mean, first_bn.running_var]).cpu().numpy(),
                               g[f"{name}_first_running"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack([deep_bn.running_mean, deep_bn.running_var]).cpu().numpy(),
                               g[f"{name}_deep_running"], rtol=2e-2, atol=2e-3)
    net.eval()
    ye = net(x)
    want = g[f"{name}_eval"]
    assert np.abs(ye.cpu().numpy() - want).max() <= TOL * np.abs(want).max()
    assert int(first_bn.num_batches_tracked) == g[f"{name}_nbt"][0]      # eval forwards leave the buffers alone


def test_layers_against_torch():
    """Each generic kernel against the torch operator it stands for, at widths that are not multiples of the tiles."""
    import ctypes
    import torch.nn.functional as F
    from noise_amd import _lib, ops
    from noise_amd._lib import ptr
    import noise_amd.ofa  # noqa: F401  (registers the entry points)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(5)
    rows = ctypes.c_int(0)
    slab = torch.empty(_lib.lib.noise_ofa_max_rows() * 2 * 1920, device=dev)

    def stats(y, C):
        coef = torch.empty(5 * C, device=dev)
        ops.bn_finalize_fwd(slab, rows.value, C, y.numel() // C, torch.ones(C, device=dev), torch.zeros(C, device=dev), None, None,
                            None, coef)
        flat = y.reshape(-1, C).double()
        torch.testing.assert_close(coef[2 * C:3 * C].double(), flat.mean(0), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(coef[3 * C:4 * C].double(), 1 / torch.sqrt(flat.var(0, unbiased=False) + 1e-5), rtol=1e-3, atol=1e-5)

    # 1x1 with sliced weights and a lazily normalised source
    # (the widths cover every channel-tile count of the kernel, the long one a grid cut to whole rounds)
    for npix, K, N, Kmax, Nmax in ((1000, 24, 144, 24, 144), (333, 136, 40, 192, 48), (70, 1152, 200, 1920, 320), (64, 8, 8, 16, 8),
                                   (300000, 32, 96, 32, 96), (5000, 64, 576, 64, 576), (2000, 48, 24, 48, 24),
                                   (2000, 32, 16, 32, 16), (90000, 16, 128, 16, 128)):
        a = torch.randn(npix, K, device=dev, generator=g)
        W = torch.randn(Nmax, Kmax, 1, 1, device=dev, generator=g) / K ** 0.5
        coef = torch.cat([torch.rand(K, device=dev, generator=g) + 0.5, torch.randn(K, device=dev, generator=g)])
        out = torch.empty(npix, N, device=dev)
        _lib.call("noise_ofa_pw", ptr(a), ptr(coef), 1, ptr(W), Kmax, ptr(out), ptr(slab), ctypes.byref(rows), npix, K, N,
                  _lib.stream())
        act = torch.clamp(a * coef[:K] + coef[K:], 0, 6)
        want = act.double() @ W[:N, :K, 0, 0].double().t()
        torch.testing.assert_close(out.double(), want, rtol=1e-4, atol=1e-4)
        stats(out, N)
    # depthwise, every kernel size / stride / direction
    for C, k, s, tr, Hi, Wi in ((24, 7, 2, False, 19, 11), (24, 7, 2, True, 9, 6), (200, 5, 1, False, 7, 9), (200, 5, 1, True, 7, 9),
                                (8, 3, 2, True, 13, 5), (1152, 3, 2, False, 6, 4), (40, 7, 1, True, 3, 3),
                                (16, 3, 2, False, 1, 37), (16, 3, 2, True, 1, 18), (96, 5, 2, False, 4, 10), (96, 5, 2, True, 4, 7),
                                (36, 3, 2, False, 5, 1), (36, 3, 2, True, 2, 1), (576, 3, 2, True, 1, 63), (576, 3, 2, False, 1, 126)):
        B = 2
        a = torch.randn(B, Hi, Wi, C, device=dev, generator=g)
        coef = torch.cat([torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g)])
        w = torch.randn(C, 1, k, k, device=dev, generator=g)
        wt = torch.empty(k * k, C, device=dev)
        _lib.call("noise_ofa_dw_filter", ptr(w), k, None, 0, None, 0, k, C, int(tr and s == 1), ptr(wt), _lib.stream())
        act = torch.clamp(a * coef[:C] + coef[C:], 0, 6).permute(0, 3, 1, 2)
        fn = F.conv_transpose2d if tr else F.conv2d
        want = fn(act.double(), w.double(), stride=s, padding=k // 2, groups=C).permute(0, 2, 3, 1)
        out = torch.empty(want.shape, device=dev)
        _lib.call("noise_ofa_dw", ptr(a), ptr(coef), 1, ptr(wt), ptr(out), ptr(slab), ctypes.byref(rows), B, Hi, Wi, C, k, s,
                  int(tr), _lib.stream())
        torch.testing.assert_close(out.double(), want, rtol=1e-4, atol=1e-4)
        stats(out, C)
    # active filter through both transform matrices
    C = 40
    w = torch.randn(C, 1, 7, 7, device=dev, generator=g)
    m75 = torch.eye(25, device=dev) + 0.1 * torch.randn(25, 25, device=dev, generator=g)
    m53 = torch.eye(9, device=dev) + 0.1 * torch.randn(9, 9, device=dev, generator=g)
    f5 = F.linear(w[:, :, 1:6, 1:6].reshape(C, 25), m75)
    f3 = F.linear(f5.view(C, 5, 5)[:, 1:4, 1:4].reshape(C, 9), m53)
    for k, want in ((7, w.view(C, 49)), (5, f5), (3, f3)):
        wt = torch.empty(k * k, C, device=dev)
        _lib.call("noise_ofa_dw_filter", ptr(w), 7, ptr(m75), 5, ptr(m53), 3, k, C, 0, ptr(wt), _lib.stream())
        torch.testing.assert_close(wt.t(), want, rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError, match="multiples of 4"):
        _lib.call("noise_ofa_pw", ptr(slab), None, 0, ptr(slab), 8, ptr(slab), ptr(slab), ctypes.byref(rows), 10, 6, 8, _lib.stream())


def test_reduced_depth_fails_like_the_reference():
    """With fewer active blocks per stage the reference's decoder runs the mirrors of a stage's LAST blocks
    (layers.py:660), which neither upsample nor change width: its skip add fails with a size-mismatch RuntimeError
    ("The size of tensor a (32) must match the size of tensor b (24) at non-singleton dimension 1" for this net)."""
    from noise_amd.ofa import OFASpectralNets
    net = OFASpectralNets(base_stage_width=[16, 8, 16, 16, 24, 24, 32, 32, 64], kernel_size_list=[3]).cuda().eval()
    net.set_active(d=[1, 1, 2, 2, 2, 2, 2, 2])
    with pytest.raises(RuntimeError, match="32 channels, the encoder skip 24"):
        net(torch.randn(1, 1, 33, 33, device="cuda"))


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_sub_networks_vs_oracle(seed):
    """Random input sizes, kernel sizes and widths on a small width table: the HIP forward against the CPU oracle
    (oracle/ofa.py, itself pinned by the reference's outputs), train mode (incl. the padding search's side effects on the
    running statistics) and eval mode."""
    import copy
    from make_golden_ofa import choose, perturb
    from noise_amd.ofa import OFASpectralNets
    from oracle import ofa as oofa
    rng = np.random.RandomState(100 + seed)
    ks_list = [[3, 5, 7], [3, 7], [5], [3, 5]][seed % 4]
    mults = [[0.5, 0.75, 1.0], [1.0], [0.5, 1.0]][seed % 3]
    torch.manual_seed(seed)
    net = OFASpectralNets(base_stage_width=[16, 8, 16, 24, 24, 32, 40, 48, 64], kernel_size_list=ks_list,
                          expand_ratio_list=[2, 3], width_mult_list=mults)
    perturb(net, seed)
    choose(net, ks=int(rng.choice(ks_list)), wid=int(rng.randint(len(mults))))
    ref = copy.deepcopy(net)
    net = net.cuda()
    B, H, W = int(rng.randint(1, 4)), int(rng.randint(20, 90)), int(rng.randint(20, 70))
    x = torch.from_numpy(rng.randn(B, 1, H, W).astype(np.float32))
    for mode in (True, True, False):
        net.train(mode)
        ref.train(mode)
        want = oofa.forward(ref, x).numpy()
        got = net(x.cuda()).cpu().numpy()
        assert got.shape == want.shape == (B, 1, H, W)
        assert np.abs(got - want).max() <= TOL * max(np.abs(want).max(), 1e-3), (mode, np.abs(got - want).max())
    assert (net.H_padding, net.W_padding) == (ref.H_padding, ref.W_padding)
    a, b = net.encoder[3].depth_conv.bn.bn, ref.encoder[3].depth_conv.bn.bn
    assert int(a.num_batches_tracked) == int(b.num_batches_tracked) > 1
    c = a.running_mean.numel() if a.running_mean.numel() < 9999 else 0
    np.testing.assert_allclose(a.running_var.cpu().numpy()[:c], b.running_var.numpy()[:c], rtol=5e-3, atol=1e-5)
