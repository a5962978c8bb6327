"""GPU parity of the waveform models' loss kernels (csrc/mrstft.hip through the C-ABI) against the numpy oracle and
the fixtures the reference wrote (train/models.py:641-720).  fp32 transforms against a double-precision oracle:
losses within 2e-5 relative; gradients within 2e-4 of the largest gradient (a bin whose two magnitudes agree to
rounding can take either sign of the L1 term, which moves single samples by that much)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = ([512, 1024, 2048], [50, 120, 240], [240, 600, 1200])


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def g5(golden_dir):
    return np.load(os.path.join(golden_dir, "g5_mrstft.npz"))


def _grad_close(got, ref, tol=2e-4):
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max() / scale
    assert err < tol, err


def test_golden_loss_and_gradient(dev, g5):
    from noise_amd import losses
    for tag in ("", "odd_"):
        x = torch.from_numpy(g5[tag + "enhanced"]).to(dev).requires_grad_(True)
        y = torch.from_numpy(g5[tag + "clean"]).to(dev)
        loss = losses.multi_resolution_stft_loss(x, y, *CFG)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g5[tag + "loss"], rtol=2e-5)
        _grad_close(x.grad.cpu().numpy(), g5[tag + "grad"])
    # per-resolution terms
    x = torch.from_numpy(g5["enhanced"]).to(dev)
    y = torch.from_numpy(g5["clean"]).to(dev)
    _, terms, _ = losses.mrstft_loss_and_grad(x, y, *CFG, want_grad=False)
    np.testing.assert_allclose(terms.cpu().numpy(), [g5["loss_0"], g5["loss_1"], g5["loss_2"]], rtol=2e-5)
    for i in range(3):
        _, _, gi = losses.mrstft_loss_and_grad(x, y, [CFG[0][i]], [CFG[1][i]], [CFG[2][i]])
        _grad_close(gi.cpu().numpy(), g5[f"grad_{i}"])


def test_golden_total_loss(dev, g5):
    from noise_amd import losses
    x = torch.from_numpy(g5["enhanced"]).to(dev).requires_grad_(True)
    y = torch.from_numpy(g5["clean"]).to(dev)
    z = torch.from_numpy(g5["total_vad_logit"]).to(dev).requires_grad_(True)
    total, l1, mr, vad = losses.waveform_loss(x, y, z, float(g5["total_lambda"]), *CFG)
    total.backward()
    np.testing.assert_allclose(total.item(), g5["total_loss"], rtol=2e-5)
    np.testing.assert_allclose(l1.item(), g5["total_l1"], rtol=2e-5)
    np.testing.assert_allclose(vad.item(), g5["total_vad"], rtol=2e-5)
    _grad_close(x.grad.cpu().numpy(), g5["total_grad"])
    assert z.grad is not None and torch.isfinite(z.grad).all()


@pytest.mark.parametrize("B,L,n_fft,hop,wl", [(2, 8000, 512, 50, 240), (1, 1100, 2048, 240, 1200), (3, 4001, 1024, 120, 600),
                                              (2, 3000, 512, 128, 512), (1, 2049, 2048, 512, 2048), (5, 777, 1024, 33, 101)])
def test_term_vs_oracle(dev, B, L, n_fft, hop, wl):
    """Single resolutions at ragged sizes: clips barely longer than the reflect padding, hops that divide nothing,
    a window as long as the transform, an odd window length, tiles that overhang the last frame."""
    from noise_amd import losses
    from oracle import mrstft as om
    rng = np.random.RandomState(L + hop)
    env = 0.5 - 0.5 * np.cos(2 * np.pi * 3.0 * np.arange(L) / L)
    y = (rng.randn(B, L) * 0.05 * env).astype(np.float32)
    x = (y + rng.randn(B, L).astype(np.float32) * 0.02).astype(np.float32)
    y[0, L // 2:] = 0.0
    x[0, 3 * L // 4:] = 0.0
    ref_loss, ref_grad = om.resolution_loss_grad(x, y, n_fft, hop, wl)
    _, terms, grad = losses.mrstft_loss_and_grad(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), [n_fft], [hop], [wl])
    np.testing.assert_allclose(terms.cpu().numpy()[0], ref_loss, rtol=2e-5)
    _grad_close(grad.cpu().numpy(), ref_grad)


def test_full_batch_properties(dev):
    """At the training batch of the reference's configs (1280 segments of 8000 samples): the loss of a signal against
    itself is exactly 0 with a zero gradient, the loss is the mean of its per-clip losses, and two runs are
    bit-identical (no atomics)."""
    from noise_amd import losses
    g = torch.Generator().manual_seed(3)
    y = (torch.randn(1280, 8000, generator=g) * 0.05).to(dev)
    x = (y + 0.01 * torch.randn(1280, 8000, generator=g).to(dev)).clamp(-1, 1)
    same, _, gsame = losses.mrstft_loss_and_grad(y, y, *CFG)
    assert same.item() == 0.0 and gsame.abs().max().item() == 0.0
    l1, t1, g1 = losses.mrstft_loss_and_grad(x, y, *CFG)
    l2, t2, g2 = losses.mrstft_loss_and_grad(x, y, *CFG)
    assert torch.equal(t1, t2) and torch.equal(g1, g2)
    per_clip = torch.stack([losses.mrstft_loss_and_grad(x[i:i + 160], y[i:i + 160], *CFG, want_grad=False)[0]
                            for i in range(0, 1280, 160)])
    np.testing.assert_allclose(per_clip.mean().item(), l1.item(), rtol=1e-9)
    # a clip's gradient does not depend on the rest of the batch except through the 1/B of the mean
    _, _, gsub = losses.mrstft_loss_and_grad(x[:160], y[:160], *CFG)
    np.testing.assert_allclose((g1[:160] * 8).cpu().numpy(), gsub.cpu().numpy(), rtol=0, atol=1e-9)


def test_smooth_l1(dev):
    from noise_amd import losses
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 1, 1, 1001, generator=g)
    y = torch.randn(7, 1, 1, 1001, generator=g)
    xr = x.clone().double().requires_grad_(True)
    ref = torch.nn.functional.smooth_l1_loss(xr, y.double(), beta=0.5)
    ref.backward()
    xd = x.to(dev).requires_grad_(True)
    got = losses.smooth_l1_loss(xd, y.to(dev), beta=0.5)
    (2.0 * got).backward()
    np.testing.assert_allclose(got.item(), ref.item(), rtol=1e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), 2.0 * xr.grad.float().numpy(), rtol=1e-6, atol=1e-9)


def test_refusals(dev):
    from noise_amd import losses
    x = torch.zeros(2, 8000, device=dev)
    with pytest.raises(RuntimeError, match="n_fft"):
        losses.mrstft_loss_and_grad(x, x, [320], [160], [320])
    with pytest.raises(RuntimeError, match="too short"):
        losses.mrstft_loss_and_grad(x[:, :1000], x[:, :1000], [2048], [240], [1200])
    with pytest.raises(ValueError):
        losses.mrstft_loss_and_grad(x, x[:, :4000], *CFG)
    with pytest.raises(RuntimeError, match="CPU"):
        losses.mrstft_loss_and_grad(x.cpu(), x.cpu(), *CFG)
