"""`STFT` with the reference's interface (train/models.py:410-548), computed by HIP kernels.

Same constructor, same three registered buffers (`forward_basis`, `inverse_basis`,
`window_sum`, so reference checkpoints load), same `forward` / `inverse` contracts:

    forward(waveform (B,1,1,L))            -> log_magnitude (B,1,F,T), phase (B,F,T)
    inverse(log_magnitude, phase)          -> waveform (B,1,1,hop*(T-1)) clamped to [-1,1]

`inverse` is differentiable w.r.t. `log_magnitude` (the only gradient the training step needs:
the phase comes from the noisy input).  There is no CPU path: tensors must live on the GPU.
"""
import os
import numpy as np
import torch
from scipy.signal import get_window

from . import _lib


def _pad_center(w, size):
    lpad = (size - len(w)) // 2
    return np.pad(w, (lpad, size - len(w) - lpad))


def window_sumsquare(window, n_frames, hop_length, win_length, n_fft):
    """Sum-of-squared-windows envelope (train/models.py:497-548), accumulated in float32 in frame
    order like the reference's loop, but vectorised over frames."""
    R = -(-n_fft // hop_length)
    win_sq = _pad_center(get_window(window, win_length, fftbins=True) ** 2, n_fft)
    win_sq = np.pad(win_sq, (0, R * hop_length - n_fft))
    Q = n_frames - 1 + R
    env = np.zeros((Q, hop_length), dtype=np.float32)
    for r in range(R - 1, -1, -1):  # increasing frame index t = q - r
        seg = win_sq[r * hop_length:(r + 1) * hop_length]
        lo, hi = r, min(Q, n_frames + r)
        env[lo:hi] = (env[lo:hi].astype(np.float64) + seg).astype(np.float32)
    return env.reshape(-1)[: n_fft + hop_length * (n_frames - 1)]


def interleaved_transposed(basis, n_fft):
    """(2F, n_fft) [Re rows ; Im rows] -> (n_fft, Rpad) with column 2f = Re f, 2f+1 = Im f, zero padded
    to a multiple of 32: the layout the frame-product kernel reads (coalesced along the rows it owns)."""
    F = n_fft // 2 + 1
    rpad = ((2 * F + 31) // 32) * 32
    out = torch.zeros(n_fft, rpad, dtype=basis.dtype, device=basis.device)
    out[:, 0:2 * F:2] = basis[:F].t()
    out[:, 1:2 * F:2] = basis[F:].t()
    return out.contiguous()


def folded(basis, n_fft):
    """(2F, n_fft) [Re rows ; Im rows] -> (n_fft/2 + 1, 2*FP): row k holds the basis at sample k, Re f in column
    f and Im f in column FP + f (FP = F rounded up to 32).  Only valid for a symmetric basis (`is_symmetric`):
    the kernel pairs sample k with n_fft - k."""
    F = n_fft // 2 + 1
    fp = ((F + 31) // 32) * 32
    out = torch.zeros(F, 2 * fp, dtype=basis.dtype, device=basis.device)
    out[:, :F] = basis[:F, :F].t()
    out[:, fp:fp + F] = basis[F:, :F].t()
    return out.contiguous()


def fft_tables(forward_basis, inverse_basis, n_fft, tol=2e-6):
    """Windows and twiddle tables for the FFT kernels, or None when the registered bases are not `window x real DFT`
    (a checkpoint with edited bases): forward_basis[f][n] = w[n] (cos, -sin)(2 pi f n / N), inverse_basis[f][n] =
    c_f ws[n] (cos, -sin)(2 pi f n / N) / N with c = 1 for the DC / Nyquist rows and 2 otherwise
    (train/models.py:413-458).  Returns (w, ws, tw1, tw2): tw1[j] = e^{-2 pi i j / (N/2)}, tw2[k] = e^{-2 pi i k / N}."""
    N, F = n_fft, n_fft // 2 + 1
    fb, ib = forward_basis.double().cpu(), inverse_basis.double().cpu()
    n = torch.arange(N, dtype=torch.float64)
    ang = 2 * np.pi * torch.arange(F, dtype=torch.float64)[:, None] * n / N
    w, ws = fb[0], ib[0] * N
    c = torch.full((F, 1), 2.0, dtype=torch.float64)
    c[0] = c[-1] = 1.0
    ok = ((fb[:F] - torch.cos(ang) * w).abs().max() <= tol * fb.abs().max() and
          (fb[F:] + torch.sin(ang) * w).abs().max() <= tol * fb.abs().max() and
          (ib[:F] - c * ws * torch.cos(ang) / N).abs().max() <= tol * ib.abs().max() and
          (ib[F:] + c * ws * torch.sin(ang) / N).abs().max() <= tol * ib.abs().max())
    if not ok:
        return None
    j = torch.arange(N // 2, dtype=torch.float64)
    k = torch.arange(N // 2 + 1, dtype=torch.float64)
    tw1 = torch.stack([torch.cos(2 * np.pi * j / (N // 2)), -torch.sin(2 * np.pi * j / (N // 2))], dim=1)
    tw2 = torch.stack([torch.cos(2 * np.pi * k / N), -torch.sin(2 * np.pi * k / N)], dim=1)
    dev = forward_basis.device
    return tuple(t.float().contiguous().to(dev) for t in (w, ws, tw1, tw2))


def is_symmetric(basis, n_fft, tol=1e-6):
    """Re rows even, Im rows odd about n_fft/2 (to `tol` of the largest entry): true whenever the window obeys
    w[n] = w[n_fft - n], as every periodic window of scipy.signal.get_window does."""
    F = n_fft // 2 + 1
    a = basis[:, 1:n_fft // 2]
    m = basis[:, n_fft // 2 + 1:].flip(-1)
    lim = tol * float(basis.abs().max())
    return bool((a[:F] - m[:F]).abs().max() <= lim and (a[F:] + m[F:]).abs().max() <= lim)


class _InverseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_magnitude, phase, stft):
        out = stft._inverse_raw(log_magnitude, phase, None)
        ctx.stft = stft
        ctx.save_for_backward(log_magnitude, phase, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lm, ph, out = ctx.saved_tensors
        return ctx.stft._inverse_bwd(grad_out, out, lm, ph), None, None


class STFT(torch.nn.Module):
    def __init__(self, max_frames, n_fft, hop_length, win_length, window):
        super().__init__()
        self.max_frames = max_frames
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.win_length = win_length
        self.window = window
        self.scale = n_fft / hop_length
        self.pad_amount = n_fft // 2
        self.cutoff = n_fft // 2 + 1

        dft = np.fft.fft(np.eye(n_fft))
        basis = np.vstack([dft[: self.cutoff].real, dft[: self.cutoff].imag])
        forward_basis = torch.from_numpy(basis).float()
        inverse_basis = torch.from_numpy(np.linalg.pinv(self.scale * basis).T).float()
        if window is not None:
            assert n_fft >= win_length
            w = torch.from_numpy(_pad_center(get_window(window, win_length, fftbins=True), n_fft)).float()
            forward_basis = forward_basis * w
            inverse_basis = inverse_basis * w
            env = window_sumsquare(window, max_frames, hop_length, win_length, n_fft)
        else:
            env = np.ones(n_fft + hop_length * (max_frames - 1), dtype=np.float32)
        env[env == 0] = 1
        window_sum = torch.from_numpy((self.scale / env).astype(np.float32))
        if window is None:  # the reference multiplies by window_sum only when a window is set
            window_sum = torch.ones_like(window_sum)

        self.register_buffer("forward_basis", forward_basis[:, None, :].contiguous())
        self.register_buffer("inverse_basis", inverse_basis[:, None, :].contiguous())
        self.register_buffer("window_sum", window_sum)
        self._derived = None  # (key, symmetric, forward layout, inverse layout) rebuilt when buffers move/change
        self.use_symmetry = True  # False forces the general kernels (any basis)
        self.use_fft = True       # False forces the dense-basis kernels
        # FFT kernels: hardware-rate log / exp / sine / cosine and a polynomial angle instead of the library functions
        # (against float64 the spectrum is within 1.3e-6 of its largest bin instead of 2.3e-7, the reconstruction within
        # 1e-7 either way; the parity bar is 1e-5).  The default of `forward` / `inverse` calls that do not say; the
        # training step always asks for the library functions.  NOISE_STFT_FAST=0 turns it off.
        self.fast_math = os.environ.get("NOISE_STFT_FAST", "1") != "0"

    # -- kernel-side layouts of the two bases, derived lazily from the registered buffers
    def _bases(self):
        """(symmetric, forward general, inverse general, forward folded, inverse folded); the folded layouts are
        None when either basis lacks the mirror symmetry (or `use_symmetry` is off)."""
        fb, ib = self.forward_basis, self.inverse_basis
        key = (fb.data_ptr(), fb._version, ib.data_ptr(), ib._version, self.use_symmetry, self.use_fft)
        if self._derived is None or self._derived[0] != key:
            f2, i2 = fb[:, 0, :], ib[:, 0, :]
            sym = self.use_symmetry and is_symmetric(f2, self.n_fft) and is_symmetric(i2, self.n_fft)
            fft = None
            # the FFT entry points need an even hop that divides n_fft with fewer than 32 overlapping frames per sample
            # (check_fft / noise_istft_fwd_fft in csrc/fft.hip); other hops stay on the dense-basis kernels
            hop = self.hop_length
            if self.use_fft and fb.is_cuda and _lib.lib.noise_fft_supported(self.n_fft) and self.n_fft % hop == 0 \
                    and hop % 2 == 0 and self.n_fft // hop < 32:
                fft = fft_tables(f2, i2, self.n_fft)
            self._derived = (key, sym, interleaved_transposed(f2, self.n_fft), interleaved_transposed(i2, self.n_fft),
                             folded(f2, self.n_fft) if sym else None, folded(i2, self.n_fft) if sym else None, fft)
        return self._derived[1:6]

    def _fft(self):
        """(analysis window, synthesis window, tw1, tw2) when the FFT kernels can serve the registered bases."""
        self._bases()
        return self._derived[6]

    def num_frames(self, L):
        return L // self.hop_length + 1

    def forward(self, waveform, return_absmax=False, reference_order=False, want_phase=True, fast_math=None):
        """`reference_order=True` takes the general dense kernel, which accumulates every bin over the n_fft samples in
        the order of the reference's conv1d, even when a faster one is available.  The kernels agree to fp32 rounding
        (the others are slightly closer to a float64 evaluation), i.e. to ~3e-5 of a log-magnitude on small bins, and
        the training step asks for the reference's order on the network input so that it reproduces the reference's
        gradients to ~1e-6 of each tensor: a different rounding of the input moves the bottleneck activations by ~3e-5,
        which flips a few ReLU6 masks, and with the handful of bottleneck pixels of a small batch one flip is ~1e-2 of a
        weight-gradient entry (measured on the 4-segment reference fixture: 4e-4 -> 8e-3 with any other input rounding).
        `reference_order="normaliser"` is the cheaper middle: the spectrum from the FFT kernel, and only the per-clip
        normaliser max|log-magnitude| (train/models.py:614-620, attained on the bin of smallest magnitude, where the
        rounding is largest) re-evaluated in the dense order on the few bins that can attain it
        (`noise_stft_fwd_fft_ref`): bit-identical normaliser at 0.12 ms instead of 0.27 ms per 1280 segments.
        `fast_math=True` (the bf16 inference pipeline; FFT kernels only) computes the log and the angle with
        hardware-rate approximations (about 2e-6 absolute) instead of the library functions."""
        _lib.require_cuda(waveform, self.forward_basis)
        fast_math = self.fast_math if fast_math is None else fast_math
        B = waveform.shape[0]
        L = waveform.shape[-1]
        wave = waveform.reshape(B, L).contiguous().float()
        T = self.num_frames(L)
        sym, basisT, _, basisF, _ = self._bases()
        sym = sym and not reference_order
        fft = self._fft()
        normaliser_only = reference_order == "normaliser" and fft is not None and return_absmax
        if reference_order and not normaliser_only:
            fft = None      # the whole transform as dense products in the reference's order
        logmag = torch.empty(B, 1, self.cutoff, T, device=wave.device, dtype=torch.float32)
        # a caller that only needs the magnitudes (the clean target of the spectral loss) skips atan2 and its store
        phase = torch.empty(B, self.cutoff, T, device=wave.device, dtype=torch.float32) if want_phase else None
        absmax = torch.empty(B, device=wave.device, dtype=torch.float32) if return_absmax else None
        nbytes = 4 * B * L + (8 if want_phase else 4) * B * self.cutoff * T
        if normaliser_only:
            _lib.call("noise_stft_fwd_fft_ref", _lib.ptr(wave), _lib.ptr(fft[0]), _lib.ptr(fft[2]), _lib.ptr(fft[3]),
                      _lib.ptr(basisT), _lib.ptr(logmag), _lib.ptr(phase), _lib.ptr(absmax), B, L, self.n_fft,
                      self.hop_length, _lib.stream(), nbytes=nbytes)
        elif fft is not None:
            _lib.call("noise_stft_fwd_fft", _lib.ptr(wave), _lib.ptr(fft[0]), _lib.ptr(fft[2]), _lib.ptr(fft[3]),
                      _lib.ptr(logmag), _lib.ptr(phase), _lib.ptr(absmax), B, L, self.n_fft, self.hop_length,
                      int(bool(fast_math)), _lib.stream(), nbytes=nbytes)
        else:
            _lib.call("noise_stft_fwd_sym" if sym else "noise_stft_fwd", _lib.ptr(wave), _lib.ptr(basisF if sym else basisT),
                      _lib.ptr(logmag), _lib.ptr(phase), _lib.ptr(absmax), B, L, self.n_fft, self.hop_length, _lib.stream(),
                      nbytes=nbytes)
        if return_absmax:
            return logmag, phase, absmax
        return logmag, phase

    def _inverse_raw(self, log_magnitude, phase, vad_logit, fast_math=None):
        _lib.require_cuda(log_magnitude, phase, self.inverse_basis)
        fast_math = self.fast_math if fast_math is None else fast_math
        B, _, F, T = log_magnitude.shape
        lm = log_magnitude.contiguous().float()
        ph = phase.contiguous().float()
        n_out = self.hop_length * (T - 1)
        assert self.n_fft + self.hop_length * (T - 1) <= self.window_sum.numel(), "more frames than max_frames"
        out = torch.empty(B, 1, 1, n_out, device=lm.device, dtype=torch.float32)
        fft = self._fft()
        if fft is not None:
            _lib.call("noise_istft_fwd_fft", _lib.ptr(lm), _lib.ptr(ph), _lib.ptr(fft[1]), _lib.ptr(fft[2]), _lib.ptr(fft[3]),
                      _lib.ptr(self.window_sum), _lib.ptr(vad_logit), _lib.ptr(out), B, T, self.n_fft, self.hop_length,
                      int(bool(fast_math)), _lib.stream(), nbytes=8 * B * F * T + 4 * B * n_out)
            return out
        _lib.call("noise_istft_fwd_sym" if self._bases()[0] else "noise_istft_fwd", _lib.ptr(lm), _lib.ptr(ph), _lib.ptr(self.inverse_basis),
                  _lib.ptr(self.window_sum), _lib.ptr(vad_logit), _lib.ptr(out), B, T, self.n_fft,
                  self.hop_length, _lib.stream(), nbytes=8 * B * F * T + 4 * B * n_out)
        return out

    def _inverse_bwd(self, grad_out, out, log_magnitude, phase, d_lm=None, accumulate=False):
        B, _, F, T = log_magnitude.shape
        sym, _, inv_basisT, _, inv_basisF = self._bases()
        g = grad_out.contiguous().float()
        if d_lm is None:
            d_lm = torch.empty_like(log_magnitude, dtype=torch.float32)
        fft = self._fft()
        if fft is not None:
            _lib.call("noise_istft_bwd_fft", _lib.ptr(g), _lib.ptr(out), _lib.ptr(log_magnitude.contiguous()),
                      _lib.ptr(phase.contiguous()), _lib.ptr(fft[1]), _lib.ptr(fft[2]), _lib.ptr(fft[3]),
                      _lib.ptr(self.window_sum), _lib.ptr(d_lm), int(accumulate), B, T, self.n_fft, self.hop_length,
                      _lib.stream(), nbytes=8 * g.numel() + (12 + 4 * int(accumulate)) * B * F * T)
            return d_lm
        _lib.call("noise_istft_bwd_sym" if sym else "noise_istft_bwd", _lib.ptr(g), _lib.ptr(out), _lib.ptr(log_magnitude.contiguous()),
                  _lib.ptr(phase.contiguous()), _lib.ptr(inv_basisF if sym else inv_basisT), _lib.ptr(self.window_sum), _lib.ptr(d_lm),
                  int(accumulate), B, T, self.n_fft, self.hop_length, _lib.stream(),
                  nbytes=8 * g.numel() + (12 + 4 * int(accumulate)) * B * F * T)
        return d_lm

    def inverse(self, log_magnitude, phase):
        if torch.is_grad_enabled() and log_magnitude.requires_grad:
            return _InverseFn.apply(log_magnitude, phase, self)
        return self._inverse_raw(log_magnitude, phase, None)
