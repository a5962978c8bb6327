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


def _stft(n_fft, hop, dev, max_frames=1500, sym=True, window="hann", win_length=None):
    """sym = "fft": the in-LDS FFT kernels (taken whenever the bases are window x DFT and n_fft is 320 or 512);
    True: the folded dense kernels (every symmetric window); False: the general dense kernels."""
    from noise_amd import STFT
    st = STFT(max_frames, n_fft, hop, win_length or n_fft, window).to(dev)
    st.use_fft = sym == "fft"
    st.use_symmetry = bool(sym)
    assert (st._fft() is not None) == (sym == "fft")
    if sym != "fft":          # (the FFT kernels do not care whether the window is symmetric)
        assert st._bases()[0] == bool(sym)
    return st


SYM = pytest.mark.parametrize("sym", ["fft", True, False], ids=["fft", "folded", "general"])


@SYM
@pytest.mark.parametrize("tag,n_fft,hop", CONFIGS)
def test_stft_golden(dev, golden_dir, tag, n_fft, hop, sym):
    g = np.load(os.path.join(golden_dir, "g1_stft.npz"))
    st = _stft(n_fft, hop, dev, sym=sym)
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
@SYM
@pytest.mark.parametrize("B,L", [(1, 8000), (3, 16000), (2, 160000), (5, 1600), (2, 12345 // 1 * 1)])
def test_stft_vs_oracle_and_roundtrip(dev, tag, n_fft, hop, B, L, sym):
    from oracle import spectral as osp
    st = _stft(n_fft, hop, dev, sym=sym)
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


@SYM
def test_istft_backward_matches_autograd(dev, sym):
    from oracle import spectral as osp
    for tag, n_fft, hop in CONFIGS:
        st = _stft(n_fft, hop, dev, sym=sym)
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


@pytest.mark.parametrize("n_fft,hop,win,window", [(320, 160, 200, "hann"), (320, 160, 320, "hamming"), (1024, 256, 1024, "hann"),
                                                 (64, 32, 64, "blackman"), (480, 160, 480, "hann")])
def test_folded_equals_general(dev, n_fft, hop, win, window):
    """Other windows, short windows padded to n_fft, the largest and smallest n_fft, n_fft/hop = 3: the folded
    kernels agree with the general ones (same arithmetic up to summation order) in all three directions."""
    a = _stft(n_fft, hop, dev, max_frames=200, sym=True, window=window, win_length=win)
    b = _stft(n_fft, hop, dev, max_frames=200, sym=False, window=window, win_length=win)
    g = torch.Generator().manual_seed(n_fft)
    w = (torch.randn(3, 1, 1, 37 * hop + 11, generator=g) * 0.3).clamp(-1, 1).to(dev)
    la, pa, ma = a.forward(w, return_absmax=True)
    lb, pb, mb = b.forward(w, return_absmax=True)
    Xa = torch.exp(la[:, 0].double()) * torch.exp(1j * pa.double())
    Xb = torch.exp(lb[:, 0].double()) * torch.exp(1j * pb.double())
    live = lb[:, 0] > np.log(1e-3) + 1e-6
    assert (Xa - Xb).abs()[live].max().item() <= 1e-5 * Xb.abs().max().item()
    np.testing.assert_allclose(ma.cpu().numpy(), mb.cpu().numpy(), rtol=1e-5)
    lm = (lb + 0.2 * torch.randn(lb.shape, generator=g).to(dev))
    outs, grads = [], []
    gout = torch.randn(3, 1, 1, hop * (lb.shape[-1] - 1), generator=g).to(dev)
    for st in (a, b):
        x = lm.clone().requires_grad_(True)
        o = st.inverse(x, pb)
        (o * gout).sum().backward()
        outs.append(o.detach().cpu().numpy())
        grads.append(x.grad.cpu().numpy())
    np.testing.assert_allclose(outs[0], outs[1], atol=1e-5)
    np.testing.assert_allclose(grads[0], grads[1], atol=1e-5 * np.abs(grads[1]).max() + 1e-6)


@pytest.mark.parametrize("n_fft,hop,win,window", [(320, 160, 200, "hann"), (320, 160, 320, "hamming"), (512, 128, 512, "blackman"),
                                                 (512, 256, 400, "hamming"), (320, 80, 320, "hann")])
def test_fft_equals_general(dev, n_fft, hop, win, window):
    """Other windows (also asymmetric after centring), short windows padded to n_fft, n_fft/hop = 2 and 4: the FFT
    kernels agree with the dense general ones in all three directions (forward, inverse, adjoint)."""
    a = _stft(n_fft, hop, dev, max_frames=200, sym="fft", window=window, win_length=win)
    b = _stft(n_fft, hop, dev, max_frames=200, sym=False, window=window, win_length=win)
    g = torch.Generator().manual_seed(n_fft + hop)
    w = (torch.randn(3, 1, 1, 37 * hop + 11, generator=g) * 0.3).clamp(-1, 1).to(dev)
    la, pa, ma = a.forward(w, return_absmax=True)
    lb, pb, mb = b.forward(w, return_absmax=True)
    Xa = torch.exp(la[:, 0].double()) * torch.exp(1j * pa.double())
    Xb = torch.exp(lb[:, 0].double()) * torch.exp(1j * pb.double())
    live = lb[:, 0] > np.log(1e-3) + 1e-6
    assert (Xa - Xb).abs()[live].max().item() <= 1e-5 * Xb.abs().max().item()
    # max |log-magnitude| is usually attained on a weak bin, where the logarithm magnifies the (absolute) 1e-5 spectrum
    # tolerance: compared a little looser than the spectrum itself
    np.testing.assert_allclose(ma.cpu().numpy(), mb.cpu().numpy(), rtol=1e-4)
    lm = (lb + 0.2 * torch.randn(lb.shape, generator=g).to(dev))
    outs, grads = [], []
    gout = torch.randn(3, 1, 1, hop * (lb.shape[-1] - 1), generator=g).to(dev)
    for st in (a, b):
        for acc in (False, True):
            x = lm.clone().requires_grad_(True)
            o = st.inverse(x, pb)
            (o * gout).sum().backward()
        outs.append(o.detach().cpu().numpy())
        grads.append(x.grad.cpu().numpy())
    np.testing.assert_allclose(outs[0], outs[1], atol=1e-5)
    np.testing.assert_allclose(grads[0], grads[1], atol=1e-5 * np.abs(grads[1]).max() + 1e-6)
    # the accumulating form of the adjoint (the training step adds the waveform-loss gradient to the spectral one)
    base = torch.randn_like(lm)
    d1, d2 = base.clone(), base.clone()
    a._inverse_bwd(gout, torch.from_numpy(outs[0]).to(dev), lm, pb, d_lm=d1, accumulate=True)
    b._inverse_bwd(gout, torch.from_numpy(outs[1]).to(dev), lm, pb, d_lm=d2, accumulate=True)
    np.testing.assert_allclose(d1.cpu().numpy(), d2.cpu().numpy(), atol=1e-5 * np.abs(grads[1]).max() + 1e-5)


@pytest.mark.parametrize("n_fft,hop", [(320, 160), (512, 128)])
def test_fft_fast_math(dev, n_fft, hop):
    """The hardware-rate log / angle / sine / cosine of the bf16 inference pipeline against the library functions of the
    same kernels: spectrum within 2e-5 of its largest bin (angle error about 2e-6 rad), reconstruction within 5e-5 on
    [-1, 1] waveforms."""
    st = _stft(n_fft, hop, dev, max_frames=200, sym="fft")
    g = torch.Generator().manual_seed(n_fft)
    w = (torch.randn(3, 1, 1, 41 * hop, generator=g) * 0.3).clamp(-1, 1).to(dev)
    le, pe, me = st.forward(w, return_absmax=True)
    lf, pf, mf = st.forward(w, return_absmax=True, fast_math=True)
    Xe = torch.exp(le[:, 0].double()) * torch.exp(1j * pe.double())
    Xf = torch.exp(lf[:, 0].double()) * torch.exp(1j * pf.double())
    assert (Xe - Xf).abs().max().item() <= 2e-5 * Xe.abs().max().item()
    np.testing.assert_allclose(mf.cpu().numpy(), me.cpu().numpy(), rtol=1e-5)
    oe = st._inverse_raw(le, pe, None)
    of = st._inverse_raw(le, pe, None, fast_math=True)
    np.testing.assert_allclose(of.cpu().numpy(), oe.cpu().numpy(), atol=5e-5)
    # end to end: analysis and synthesis both fast still reconstruct the clip
    rec = st._inverse_raw(lf, pf, None, fast_math=True)
    np.testing.assert_allclose(rec.cpu().numpy()[..., hop:-hop], w.cpu().numpy()[..., hop:-hop], atol=1e-4)


@pytest.mark.parametrize("tag,n_fft,hop", CONFIGS)
def test_reference_order_normaliser(dev, tag, n_fft, hop):
    """The per-clip normaliser max|log-magnitude| (models.py:614-620) is attained on the bin of smallest magnitude, where
    fp32 rounding depends on the summation order.  `reference_order="normaliser"` takes the spectrum from the FFT kernel and
    re-evaluates the bins that can attain the max in the k-ordered chain of the dense kernel: the normaliser must equal
    the dense kernel's bit for bit (silent clips, clips near the floor and half-silent clips included), while the
    spectrum is the plain FFT kernel's."""
    from noise_amd.data import synth_batch
    st = _stft(n_fft, hop, dev, sym="fft")
    for B, L, seed in ((300, 8000, 1), (5, 160000, 2), (7, 12345, 3)):
        x = synth_batch(B, L, dev, seed=seed)[0]
        x[0] = 0
        x[1] = x[1] * 1e-3
        x[2, ..., : L // 2] = 0
        x[3] = x[3] * 1e-2
        ld, pd, ad = st.forward(x, return_absmax=True, reference_order=True)
        lf, pf, af = st.forward(x, return_absmax=True, reference_order="normaliser")
        lp, pp, ap = st.forward(x, return_absmax=True, fast_math=False)
        assert torch.equal(af, ad)
        assert torch.equal(lf, lp) and torch.equal(pf, pp)
        assert (ap != ad).any() or L != 8000      # the plain FFT normaliser does differ on some segments
        assert float(af[0]) == float(np.float32(abs(ld[0].min().item())))


def test_asymmetric_basis_takes_general_kernels(dev):
    from noise_amd import STFT
    st = _stft(320, 160, dev, max_frames=100)
    with torch.no_grad():
        st.forward_basis[3, 0, 17] += 0.01
    assert st._bases()[0] is False
    st.use_fft = True
    assert st._fft() is None          # an edited basis is no longer window x DFT: dense kernels
    # a periodic Hamming window shorter than n_fft: its first sample (0.08) has no mirror partner after centring
    assert STFT(100, 320, 160, 200, "hamming").to(dev)._bases()[0] is False


@pytest.mark.parametrize("hop", [10])
def test_small_hops_take_the_dense_kernels(dev, hop):
    """The FFT kernels need an even hop with fewer than 32 overlapping frames per sample; for other divisors of n_fft
    the analysis is served by the dense-basis kernels instead of failing inside the FFT entry points."""
    from oracle import spectral as osp
    from noise_amd import STFT
    st = STFT(400, 320, hop, 320, "hann").to(dev)
    st.use_fft = True
    assert st._fft() is None
    w = (torch.randn(2, 1, 1, 320 * 4, generator=torch.Generator().manual_seed(hop)) * 0.1).clamp(-1, 1)
    fwd, inv, wsum = osp.stft_buffers(400, 320, hop)
    lm_ref, ph_ref = osp.stft_forward(w, fwd, 320, hop)
    lm, ph = st.forward(w.to(dev))
    X = torch.exp(lm[:, 0].cpu().double()) * torch.exp(1j * ph.cpu().double())
    Xr = torch.exp(lm_ref[:, 0].double()) * torch.exp(1j * ph_ref.double())
    floor = lm_ref[:, 0] > np.log(1e-3) + 1e-6
    assert (X - Xr).abs()[floor].max().item() <= 1e-5 * Xr.abs().max().item()
    # the overlap-add kernels (dense and FFT alike) gather at most 31 frames per sample: these hops have 32 and 64
    with pytest.raises(RuntimeError, match="n_fft/hop"):
        st.inverse(lm, ph)


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
