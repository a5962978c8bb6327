"""Training loss of the waveform models on the GPU (train/models.py:641-720, `Base._shared_step`,
`Base.stft_mag`, `Base.multi_resolution_stft_loss`): smooth-L1 on the waveform + lambda * multi-resolution STFT
log-magnitude L1 + VAD binary cross-entropy.  Every value and the gradient with respect to the enhanced waveform come
from the HIP kernels of csrc/mrstft.hip through the C-ABI; the forward call already computes the gradient (one pass
over the frames serves both), autograd only scales it by the incoming gradient."""
import ctypes
import math

import torch

from . import _lib
from ._lib import _F, _I, _L, _P, ptr

_lib.register({
    "noise_mrstft_supported": [_I],
    "noise_mrstft_partials": [_I, _I, _I, _I],
    "noise_mrstft_term": [_P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "noise_smooth_l1_blocks": [],
    "noise_smooth_l1": [_P, _P, _L, _F, _F, _P, _P, _I, _P, _I, _P],
    "noise_vad_label": [_P, _P, _I, _I, _I, _P],
})

_TABLES = {}


def _tables(n_fft, win_length, device):
    """Window of torch.stft (periodic Hann of win_length samples centred in n_fft) and the two twiddle tables of the
    packed real transform: e^{-2 pi i j / (n_fft/2)}, j < n_fft/2, and e^{-2 pi i k / n_fft}, k <= n_fft/2."""
    key = (n_fft, win_length, str(device))
    if key not in _TABLES:
        n = torch.arange(win_length, dtype=torch.float64)
        w = torch.zeros(n_fft, dtype=torch.float64)
        left = (n_fft - win_length) // 2
        w[left:left + win_length] = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / win_length)
        n2 = n_fft // 2
        a1 = -2.0 * math.pi * torch.arange(n2, dtype=torch.float64) / n2
        a2 = -2.0 * math.pi * torch.arange(n2 + 1, dtype=torch.float64) / n_fft
        tw1 = torch.stack([torch.cos(a1), torch.sin(a1)], dim=1)
        tw2 = torch.stack([torch.cos(a2), torch.sin(a2)], dim=1)
        _TABLES[key] = tuple(t.float().contiguous().to(device) for t in (w, tw1, tw2))
    return _TABLES[key]


class _Workspace:
    """Per-(device, batch shape) scratch: frame gradients of the largest resolution, loss partials, loss scalars."""
    _cache = {}

    @classmethod
    def get(cls, B, L, fft_sizes, hop_lengths, win_lengths, device):
        key = (B, L, tuple(fft_sizes), tuple(hop_lengths), tuple(win_lengths), str(device))
        ws = cls._cache.get(key)
        if ws is None:
            if len(cls._cache) >= 4:
                cls._cache.clear()
            frames = max(B * (L // h + 1) * w for h, w in zip(hop_lengths, win_lengths))
            parts = max([_lib.lib.noise_mrstft_partials(B, L, n, h) for n, h in zip(fft_sizes, hop_lengths)]
                        + [_lib.lib.noise_smooth_l1_blocks()])
            ws = cls._cache[key] = (torch.empty(frames, device=device), torch.empty(parts, device=device, dtype=torch.float64))
        return ws


def _check_pair(x, y):
    _lib.require_cuda(x, y)
    if x.dtype != torch.float32 or y.dtype != torch.float32:
        raise TypeError("waveform losses compute in fp32")
    if x.numel() != y.numel() or x.shape[-1] != y.shape[-1]:
        raise ValueError(f"enhanced {tuple(x.shape)} and clean {tuple(y.shape)} differ")


def mrstft_loss_and_grad(x, y, fft_sizes, hop_lengths, win_lengths, want_grad=True):
    """sum over resolutions of mean |log mag(y) - log mag(x)| (models.py:700-720) as a 0-d float64 tensor, the
    per-resolution terms (float64 [R]) and d loss / d x (or None)."""
    _check_pair(x, y)
    if not (len(fft_sizes) == len(hop_lengths) == len(win_lengths)) or len(fft_sizes) == 0:
        raise ValueError("fft_sizes, hop_lengths and win_lengths must be non-empty lists of one length")
    L = x.shape[-1]
    x2, y2 = x.contiguous().view(-1, L), y.contiguous().view(-1, L)
    B = x2.shape[0]
    frame_grad, partial = _Workspace.get(B, L, fft_sizes, hop_lengths, win_lengths, x.device)
    terms = torch.empty(len(fft_sizes), device=x.device, dtype=torch.float64)
    grad = torch.empty_like(x2) if want_grad else None
    st = _lib.stream()
    for i, (n, h, w) in enumerate(zip(fft_sizes, hop_lengths, win_lengths)):
        win, tw1, tw2 = _tables(n, w, x.device)
        _lib.call("noise_mrstft_term", ptr(x2), ptr(y2), ptr(win), ptr(tw1), ptr(tw2), 1.0,
                  ptr(frame_grad) if want_grad else None, ptr(partial), ctypes_offset(terms, i), 0,
                  ptr(grad), int(i > 0), B, L, n, h, w, st)
    return terms.sum(), terms, (grad.view(x.shape) if want_grad else None)


def ctypes_offset(t, index):
    """Device pointer of element `index` of a contiguous tensor."""
    return ctypes.c_void_p(t.data_ptr() + index * t.element_size())


def smooth_l1_and_grad(x, y, beta, want_grad=True):
    """torch.nn.functional.smooth_l1_loss(x, y, beta=beta) (mean) as a 0-d float64 tensor, and its gradient w.r.t. x."""
    _check_pair(x, y)
    xc, yc = x.contiguous(), y.contiguous()
    partial = torch.empty(_lib.lib.noise_smooth_l1_blocks(), device=x.device, dtype=torch.float64)
    loss = torch.empty(1, device=x.device, dtype=torch.float64)
    grad = torch.empty_like(xc) if want_grad else None
    _lib.call("noise_smooth_l1", ptr(xc), ptr(yc), xc.numel(), float(beta), 1.0, ptr(partial), ptr(loss), 0, ptr(grad), 0,
              _lib.stream())
    return loss[0], grad


class _PrecomputedGrad(torch.autograd.Function):
    """A scalar whose gradient with respect to `x` was computed together with its value."""

    @staticmethod
    def forward(ctx, x, value, grad):
        ctx.save_for_backward(grad)
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.to(grad.dtype), None, None


def multi_resolution_stft_loss(x, y, fft_sizes=(512, 1024, 2048), hop_lengths=(50, 120, 240), win_lengths=(240, 600, 1200)):
    """Drop-in for `Base.multi_resolution_stft_loss(x, y)` (models.py:700-720; defaults models.py:1109-1112):
    a float32 scalar that back-propagates into x."""
    need = x.requires_grad and torch.is_grad_enabled()
    total, _, grad = mrstft_loss_and_grad(x.detach(), y.detach(), fft_sizes, hop_lengths, win_lengths, want_grad=need)
    total = total.float()
    return _PrecomputedGrad.apply(x, total, grad) if need else total


def smooth_l1_loss(x, y, beta=0.5):
    """Drop-in for F.smooth_l1_loss(x, y, beta=beta) as models.py:657 calls it."""
    need = x.requires_grad and torch.is_grad_enabled()
    val, grad = smooth_l1_and_grad(x.detach(), y.detach(), beta, want_grad=need)
    val = val.float()
    return _PrecomputedGrad.apply(x, val, grad) if need else val


def waveform_loss(enhanced, clean, vad_logit, mr_stft_lambda=0.5, fft_sizes=(512, 1024, 2048), hop_lengths=(50, 120, 240),
                  win_lengths=(240, 600, 1200), min_active=160):
    """The loss of `Base._shared_step` (models.py:650-665): returns (loss, waveform_loss, spectral_loss, vad_loss) as
    the reference logs them; `loss` back-propagates into `enhanced` and `vad_logit`."""
    _check_pair(enhanced, clean)
    L = clean.shape[-1]
    B = clean.numel() // L
    label = torch.empty(B, device=clean.device)
    _lib.call("noise_vad_label", ptr(clean.detach().contiguous()), ptr(label), B, L, int(min_active), _lib.stream())
    vad_loss = torch.nn.functional.binary_cross_entropy_with_logits(vad_logit.reshape(B), label)
    l1 = smooth_l1_loss(enhanced, clean, beta=0.5)
    mr = mr_stft_lambda * multi_resolution_stft_loss(enhanced, clean, fft_sizes, hop_lengths, win_lengths)
    return l1 + mr + vad_loss, l1, mr, vad_loss
