"""GPU parity of the loss kernels (incl. the sorted cross-pair waveform L1) and the fused AdamW."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,L", [(3, 100), (17, 1000), (64, 777), (100, 800), (129, 50), (257, 64), (300, 61), (600, 24),
                                 (1280, 96), (1280, 101), (2048, 40), (2100, 33)])
def test_wave_pair_l1(dev, B, L):
    """mean over all (i,j,t) of |w_i[t] - c_j[t]| and its gradient, brute-force (small / huge B) and sorted paths;
    ties (clamped +-1 samples, silent clips) must give sign 0 exactly like torch's L1 subgradient."""
    from noise_amd import _lib, engine  # noqa: F401
    from noise_amd._lib import ptr
    g = torch.Generator().manual_seed(B * 7 + L)
    w = (torch.randn(B, L, generator=g) * 0.7).clamp(-1, 1)
    c = (torch.randn(B, L, generator=g) * 0.7).clamp(-1, 1)
    c[B // 2] = 0.0
    w[0, : L // 2] = 0.0
    w[-1] = c[0]                                  # exact ties against a non-trivial row
    wr = w.double().requires_grad_(True)
    ref = torch.zeros((), dtype=torch.float64)
    for j0 in range(0, B, 64):                    # chunked (B,B,L) broadcast
        ref = ref + (wr[:, None, :] - c.double()[None, j0:j0 + 64, :]).abs().sum()
    ref = ref / (B * B * L)
    ref.backward()
    wd, cd = w.to(dev), c.to(dev)
    d_w = torch.empty(B, L, device=dev)
    partial = torch.zeros((L + 1) // 2 + 8, dtype=torch.float64, device=dev)
    n = ctypes.c_int(0)
    _lib.call("noise_wave_pair_l1", ptr(wd), ptr(cd), ptr(d_w), ptr(partial), ctypes.byref(n), B, L, _lib.stream())
    got = partial[: n.value].sum().item() / (B * B * L)
    assert abs(got - ref.item()) <= 1e-6 * abs(ref.item())
    np.testing.assert_allclose(d_w.cpu().numpy(), wr.grad.float().numpy(), rtol=1e-5, atol=1e-12)


def test_spectral_vad_finalize(dev):
    from noise_amd import _lib, engine  # noqa: F401
    from noise_amd._lib import ptr
    g = torch.Generator().manual_seed(1)
    B, H, W, L = 5, 161, 51, 8000
    enh = torch.randn(B, H, W, generator=g)
    clm = torch.randn(B, H, W, generator=g)
    clm[0, :3] = enh[0, :3]                      # exact zeros of the difference
    clean = torch.randn(B, L, generator=g) * 0.1
    clean[1] = 0
    clean[2, 159:] = 0                           # 159 non-zero samples -> label 0
    clean[3, 160:] = 0                           # 160 -> label 1
    vad = torch.randn(B, generator=g) * 3
    enh_r = enh.clone().requires_grad_(True)
    vad_r = vad.clone().requires_grad_(True)
    spec = torch.nn.functional.l1_loss(enh_r, clm)
    label = ((clean.abs() > 0).sum(-1) >= 160).float()
    bce = torch.nn.functional.binary_cross_entropy_with_logits(vad_r, label)
    (spec + bce).backward()
    d_enh = torch.empty(B, H, W, device=dev)
    sp = torch.zeros(1024, dtype=torch.float64, device=dev)
    wp = torch.zeros(8, dtype=torch.float64, device=dev)
    lab = torch.empty(B, device=dev)
    d_vad = torch.empty(B, device=dev)
    out4 = torch.empty(4, device=dev)
    n = ctypes.c_int(0)
    st = _lib.stream()
    clean_d, enh_d, clm_d, vad_d = clean.to(dev), enh.to(dev), clm.to(dev), vad.to(dev)   # keep the device copies alive
    _lib.call("noise_vad_label", ptr(clean_d), ptr(lab), B, L, 160, st)
    assert lab.cpu().tolist() == label.tolist() == [1.0, 0.0, 0.0, 1.0, 1.0]
    _lib.call("noise_spectral_l1", ptr(enh_d), ptr(clm_d), ptr(d_enh), ptr(sp), ctypes.byref(n), B * H * W, st)
    _lib.call("noise_loss_finalize", ptr(sp), n.value, B * H * W, ptr(wp), 1, B, L, ptr(vad_d), ptr(lab), ptr(d_vad),
              ptr(out4), st)
    o = out4.cpu().numpy()
    np.testing.assert_allclose(o[3], spec.item(), rtol=1e-6)
    np.testing.assert_allclose(o[1], bce.item(), rtol=1e-6)
    np.testing.assert_allclose(o[0], spec.item() + bce.item(), rtol=1e-6)
    np.testing.assert_allclose(d_enh.cpu().numpy(), enh_r.grad.numpy(), rtol=1e-6, atol=0)
    np.testing.assert_allclose(d_vad.cpu().numpy(), vad_r.grad.numpy(), rtol=1e-5, atol=1e-8)


def test_fused_adamw_matches_torch(dev):
    from noise_amd import _lib, engine  # noqa: F401
    from noise_amd._lib import ptr
    g = torch.Generator().manual_seed(2)
    n = 76146
    p0 = torch.randn(n, generator=g)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    p = p0.to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step = torch.zeros((), dtype=torch.int64, device=dev)
    for it in range(5):
        grad = torch.randn(n, generator=g) * 10 ** float(torch.randint(-6, 1, (1,), generator=g))
        pt.grad = grad.clone()
        opt.step()
        # a summed 4-rank gradient with grad_scale = 1/4 must equal the mean gradient
        g4 = (grad * 4).to(dev)
        _lib.call("noise_adamw_step", ptr(p), ptr(g4), ptr(m), ptr(v), ptr(step), n, 1e-4, 0.9, 0.999, 1e-8,
                  1e-2, 0.25, _lib.stream())
        np.testing.assert_allclose(p.cpu().numpy(), pt.detach().numpy(), rtol=2e-6, atol=2e-7)
    assert int(step.item()) == 5
