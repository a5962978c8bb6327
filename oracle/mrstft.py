"""CPU oracle (test infrastructure, never the product path) for the waveform models' multi-resolution STFT loss.

Restates train/models.py:687-720 (`Base.stft_mag`, `Base.multi_resolution_stft_loss`) and the loss composition of
train/models.py:650-665 (`Base._shared_step`) in numpy, double precision, with the analytic gradient with respect to
the enhanced waveform.  `torch.stft` as the reference calls it: centred frames over a reflect-padded signal
(n_fft // 2 samples each side), a periodic Hann window of `win_length` samples zero-padded to n_fft on both sides,
one-sided spectrum, no normalisation.  Pinned by tests/golden/g5_mrstft.npz, written by the reference itself
(tests/golden/make_golden.py)."""
import math

import numpy as np

CLAMP = math.sqrt(1e-7)          # models.py:698


def padded_window(n_fft, win_length):
    """hann_window(win_length) (periodic) centred in n_fft samples, as torch.stft pads a short window."""
    n = np.arange(win_length, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
    left = (n_fft - win_length) // 2
    out = np.zeros(n_fft, dtype=np.float64)
    out[left:left + win_length] = w
    return out


def _frames(x, n_fft, hop):
    """[B][L] -> reflect-padded signal [B][L + n_fft] and its frame view [B][T][n_fft], T = 1 + L // hop."""
    pad = n_fft // 2
    xp = np.pad(np.asarray(x, dtype=np.float64), ((0, 0), (pad, pad)), mode="reflect")
    T = 1 + x.shape[1] // hop
    idx = np.arange(T)[:, None] * hop + np.arange(n_fft)[None, :]
    return xp, xp[:, idx]


def stft_mag(x, n_fft, hop, win_length):
    """models.py:687-698: |STFT| clamped from below, [B][T][F]."""
    _, fr = _frames(x, n_fft, hop)
    spec = np.fft.rfft(fr * padded_window(n_fft, win_length), axis=-1)
    return np.maximum(np.abs(spec), CLAMP)


def resolution_loss_grad(x, y, n_fft, hop, win_length):
    """One term of models.py:700-720, mean |log y_mag - log x_mag|, and its gradient with respect to x ([B][L])."""
    B, L = x.shape
    w = padded_window(n_fft, win_length)
    _, fx = _frames(x, n_fft, hop)
    _, fy = _frames(y, n_fft, hop)
    X = np.fft.rfft(fx * w, axis=-1)
    Y = np.fft.rfft(fy * w, axis=-1)
    ax, ay = np.abs(X), np.abs(Y)
    mx, my = np.maximum(ax, CLAMP), np.maximum(ay, CLAMP)
    diff = np.log(my) - np.log(mx)
    count = diff.size
    loss = np.abs(diff).mean()
    # d loss / d|X| = -sign(diff) / (count * mx) where the clamp is inactive
    a = np.where(ax > CLAMP, -np.sign(diff) / (count * mx), 0.0)
    G = a * X / np.where(ax > 0, ax, 1.0)
    # sum_k Re(G_k e^{+i theta k n}) over the one-sided bins = n_fft * irfft(H) with the inner bins halved
    H = 0.5 * G
    H[..., 0] = G[..., 0].real
    H[..., -1] = G[..., -1].real
    gfr = np.fft.irfft(H, n=n_fft, axis=-1) * n_fft * w            # [B][T][n_fft]
    pad = n_fft // 2
    gp = np.zeros((B, L + n_fft), dtype=np.float64)
    T = gfr.shape[1]
    idx = (np.arange(T)[:, None] * hop + np.arange(n_fft)[None, :]).reshape(-1)
    for b in range(B):
        np.add.at(gp[b], idx, gfr[b].reshape(-1))
    g = gp[:, pad:pad + L].copy()
    # adjoint of the reflect padding: padded sample p < pad mirrors sample pad - p, p >= pad + L mirrors 2(L-1) - (p - pad)
    g[:, 1:pad + 1] += gp[:, :pad][:, ::-1]
    g[:, L - 1 - pad:L - 1] += gp[:, pad + L:][:, ::-1]
    return loss, g


def multi_resolution_stft_loss(x, y, fft_sizes, hop_lengths, win_lengths, want_grad=False):
    """models.py:700-720.  x, y: [..., L] (flattened to [-1][L] as the reference's `.view(-1, L)` does)."""
    shape = np.shape(x)
    x2 = np.asarray(x, dtype=np.float64).reshape(-1, shape[-1])
    y2 = np.asarray(y, dtype=np.float64).reshape(-1, shape[-1])
    total, grad = 0.0, np.zeros_like(x2)
    parts = []
    for n, h, w in zip(fft_sizes, hop_lengths, win_lengths):
        li, gi = resolution_loss_grad(x2, y2, n, h, w)
        total += li
        grad += gi
        parts.append(li)
    if want_grad:
        return total, grad.reshape(shape), parts
    return total


def smooth_l1(x, y, beta):
    """torch.nn.functional.smooth_l1_loss, mean reduction (models.py:657), and its gradient with respect to x."""
    d = np.asarray(x, dtype=np.float64) - np.asarray(y, dtype=np.float64)
    ad = np.abs(d)
    val = np.where(ad < beta, 0.5 * d * d / beta, ad - 0.5 * beta)
    g = np.where(ad < beta, d / beta, np.sign(d)) / d.size
    return val.mean(), g


def waveform_loss(enhanced, clean, vad_logit, mr_stft_lambda, fft_sizes, hop_lengths, win_lengths, min_active=160):
    """models.py:650-665: smooth-L1 + lambda * multi-resolution STFT loss + VAD binary cross-entropy.
    Returns (total, dict of parts, d total / d enhanced, d total / d vad_logit)."""
    z = np.asarray(vad_logit, dtype=np.float64).reshape(-1)
    label = ((np.abs(np.asarray(clean)) > 0).sum(axis=-1) >= min_active).astype(np.float64).reshape(-1)
    bce = np.maximum(z, 0) - z * label + np.log1p(np.exp(-np.abs(z)))
    vad = bce.mean()
    d_vad = (1.0 / (1.0 + np.exp(-z)) - label) / z.size
    l1, g1 = smooth_l1(enhanced, clean, 0.5)
    mr, gm, _ = multi_resolution_stft_loss(enhanced, clean, fft_sizes, hop_lengths, win_lengths, want_grad=True)
    total = l1 + mr_stft_lambda * mr + vad
    return total, {"l1": l1, "mrstft": mr_stft_lambda * mr, "vad": vad}, g1 + mr_stft_lambda * gm, d_vad
