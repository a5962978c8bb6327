"""GPU parity: HIP STFT / iSTFT against the CPU oracle and the golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


CONFIGS = [("A", 320, 160), ("B", 512, 128)]


def _stft(n_fft, hop, dev, max_frames=1500):
    from noise_amd import STFT
    return STFT(max_frames, n_fft, hop, n_fft, "hann").to(dev)


@pytest.mark.parametrize("tag,n_fft,hop", CONFIGS)
def test_stft_golden(dev, golden_dir, tag, n_fft, hop):
    g = np.load(os.path.join(golden_dir, "g1_stft.npz"))
    st = _stft(n_fft, hop, dev)
    for s in ("seg", "long"):
        w = torch.from_numpy(g[f"{tag}_{s}_wave"]).to(dev)
        lm, ph, amax = st.forward(w, return_absmax=True)
        ref_lm = g[f"{tag}_{s}_logmag"]
        # tolerance: 1e-5 relative on the spectrum (|d log|X|| = relative error of |X|)
        np.testing.assert_allclose(lm.cpu().numpy(), ref_lm, atol=1e-5 * np.abs(ref_lm).max() + 1e-5)
        d = np.abs(np.exp(1j * ph.cpu().numpy()) - np.exp(1j * g[f"{tag}_{s}_phase"]))
        strong = ref_lm[:, 0] > np.log(1e-2)
        assert d[strong].max() < 1e-3
        np.testing.assert_allclose(amax.cpu().numpy(), np.abs(ref_lm).reshape(ref_lm.shape[0], -1).max(1), rtol=1e-5)
        rec = st.inverse(torch.from_numpy(ref_lm).to(dev), torch.from_numpy(g[f"{tag}_{s}_phase"]).to(dev))
        np.testing.assert_allclose(rec.cpu().numpy(), g[f"{tag}_{s}_recon"], atol=1e-5)


@pytest.mark.parametrize("tag,n_fft,hop", CONFIGS)
@pytest.mark.parametrize("B,L", [(1, 8000), (3, 16000), (2, 160000), (5, 1600), (2, 12345 // 1 * 1)])
def test_stft_vs_oracle_and_roundtrip(dev, tag, n_fft, hop, B, L):
    from oracle import spectral as osp
    st = _stft(n_fft, hop, dev)
    g = torch.Generator().manual_seed(L + B)
    w = (torch.randn(B, 1, 1, L, generator=g) * 0.1).clamp(-1, 1)
    fwd, inv, wsum = osp.stft_buffers(1500, n_fft, hop)
    lm_ref, ph_ref = osp.stft_forward(w, fwd, n_fft, hop)
    lm, ph = st.forward(w.to(dev))
    assert lm.shape == lm_ref.shape and ph.shape == ph_ref.shape
    # 1e-5 relative on the complex spectrum
    X = torch.exp(lm[:, 0].cpu().double()) * torch.exp(1j * ph.cpu().double())
    Xr = torch.exp(lm_ref[:, 0].double()) * torch.exp(1j * ph_ref.double())
    floor = lm_ref[:, 0] > np.log(1e-3) + 1e-6     # bins at the 1e-3 magnitude floor carry no phase information
    err = (X - Xr).abs()[floor].max().item()
    assert err <= 1e-5 * Xr.abs().max().item(), err
    rec = st.inverse(lm, ph).cpu()
    rec_ref = osp.stft_inverse(lm_ref, ph_ref, inv, wsum, n_fft, hop)
    np.testing.assert_allclose(rec.numpy(), rec_ref.numpy(), atol=1e-5)
    n = rec.shape[-1]
    # size-independent property: iSTFT(STFT(x)) == x away from the (possibly truncated) tail
    np.testing.assert_allclose(rec.numpy()[..., : n - n_fft], w.numpy()[..., : n - n_fft], atol=1e-5)


def test_istft_backward_matches_autograd(dev):
    from oracle import spectral as osp
    for tag, n_fft, hop in CONFIGS:
        st = _stft(n_fft, hop, dev)
        fwd, inv, wsum = osp.stft_buffers(1500, n_fft, hop)
        g = torch.Generator().manual_seed(5)
        w = (torch.randn(3, 1, 1, 16 * hop * 5, generator=g) * 0.4).clamp(-1, 1)
        lm, ph = osp.stft_forward(w, fwd, n_fft, hop)
        lm = (lm + 0.3 * torch.randn(lm.shape, generator=g)).requires_grad_(True)   # pushes some samples past the clamp
        out = osp.stft_inverse(lm, ph, inv, wsum, n_fft, hop)
        gout = torch.randn(out.shape, generator=g)
        (out * gout).sum().backward()
        lm_d = lm.detach().to(dev).requires_grad_(True)
        out_d = st.inverse(lm_d, ph.to(dev))
        assert (out_d.detach().abs() >= 1).any(), "test should exercise the clamp mask"
        (out_d * gout.to(dev)).sum().backward()
        ref = lm.grad.numpy()
        np.testing.assert_allclose(lm_d.grad.cpu().numpy(), ref, atol=1e-5 * np.abs(ref).max() + 1e-6)


def test_stft_rejects_bad_arguments(dev):
    from noise_amd import STFT
    st = STFT(100, 320, 160, 320, "hann")
    with pytest.raises(RuntimeError):
        st.forward(torch.zeros(1, 1, 1, 8000))          # CPU tensor: no fallback
    st = st.to(dev)
    with pytest.raises(RuntimeError):
        st.forward(torch.zeros(1, 1, 1, 100, device=dev))  # shorter than the reflect padding
    bad = STFT(100, 320, 100, 320, "hann").to(dev)          # hop does not divide n_fft
    with pytest.raises(RuntimeError):
        bad.forward(torch.zeros(1, 1, 1, 8000, device=dev))
