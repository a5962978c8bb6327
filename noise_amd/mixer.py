"""Per-utterance SNR mixing on the GPU, mirroring process/audiolib.py's call surface.

`segmental_snr_mixer(params, clean, noise, snr, ...)` keeps the reference's signature and return
tuple for one utterance (process/audiolib.py:176-216); `mix_batch` is the batched form the
sharded synthesis driver uses (one workgroup per utterance, float64 like the reference).
The random output level the reference draws inside the mixer with the global numpy RNG is drawn
the same way on the host (`np.random.randint(lower, upper)`), or passed in explicitly.
"""
import numpy as np
import torch

from . import _lib


def mix_batch(clean, noise, snr, level, target_level=-25.0, clipping_threshold=0.99, with_peaks=False):
    """clean, noise: (U, L) float64 CUDA tensors; snr, level: (U,) int32 CUDA tensors.
    Returns (clean_out, noise_out, noisy_out, level_out); with_peaks adds (peaks, noisy_f32): per utterance the (U, 4)
    values max|clean_out|, max|noise_out|, max|noisy_out| (what `is_clipped`, audiolib.py:18-19, decides on) and the sum
    of the mix (a checksum), and a float32 copy of the mix, all written by the mixing pass itself."""
    _lib.require_cuda(clean, noise, snr, level)
    assert clean.dtype == torch.float64 and noise.dtype == torch.float64
    assert clean.shape == noise.shape and clean.dim() == 2
    U, L = clean.shape
    clean, noise = clean.contiguous(), noise.contiguous()
    snr = snr.to(torch.int32).contiguous()
    level = level.to(torch.int32).contiguous()
    c = torch.empty_like(clean)
    n = torch.empty_like(clean)
    y = torch.empty_like(clean)
    lv = torch.empty(U, dtype=torch.int32, device=clean.device)
    peaks = torch.empty(U, 4, dtype=torch.float64, device=clean.device) if with_peaks else None
    y32 = torch.empty(U, L, dtype=torch.float32, device=clean.device) if with_peaks else None
    _lib.call("noise_snr_mix_peaks", _lib.ptr(clean), _lib.ptr(noise), _lib.ptr(snr), _lib.ptr(level), _lib.ptr(c),
              _lib.ptr(n), _lib.ptr(y), _lib.ptr(lv), _lib.ptr(peaks), _lib.ptr(y32), U, L, float(target_level),
              float(clipping_threshold), _lib.stream(), nbytes=U * L * (56 + (4 if with_peaks else 0)))
    return (c, n, y, lv, peaks, y32) if with_peaks else (c, n, y, lv)


def activity_batch(audio, fs=16000, energy_thresh=0.13, target_level=-25.0):
    """audio (U, L) float64 CUDA -> fraction of 50 ms frames judged active, (U,) float64."""
    _lib.require_cuda(audio)
    assert audio.dtype == torch.float64 and audio.dim() == 2
    U, L = audio.shape
    out = torch.empty(U, dtype=torch.float64, device=audio.device)
    _lib.call("noise_activity", _lib.ptr(audio.contiguous()), _lib.ptr(out), U, L, int(fs), float(energy_thresh),
              float(target_level), _lib.stream())
    return out


def _to_dev(x, device):
    return torch.as_tensor(np.asarray(x, dtype=np.float64), device=device)


def segmental_snr_mixer(params, clean, noise, snr, target_level=-25, clipping_threshold=0.99, device="cuda"):
    """Drop-in for audiolib.segmental_snr_mixer: numpy in, numpy out, one utterance."""
    clean = np.asarray(clean, dtype=np.float64)
    noise = np.asarray(noise, dtype=np.float64)
    if len(clean) > len(noise):
        noise = np.append(noise, np.zeros(len(clean) - len(noise)))
    else:
        clean = np.append(clean, np.zeros(len(noise) - len(clean)))
    level = np.random.randint(params["target_level_lower"], params["target_level_upper"])
    c, n, y, lv = mix_batch(_to_dev(clean, device)[None], _to_dev(noise, device)[None],
                            torch.tensor([int(snr)], dtype=torch.int32, device=device),
                            torch.tensor([int(level)], dtype=torch.int32, device=device),
                            target_level, clipping_threshold)
    return c[0].cpu().numpy(), n[0].cpu().numpy(), y[0].cpu().numpy(), int(lv.item())


def activitydetector(audio, fs=16000, energy_thresh=0.13, target_level=-25, device="cuda"):
    """Drop-in for audiolib.activitydetector (process/audiolib.py:251-289)."""
    return float(activity_batch(_to_dev(audio, device)[None], fs, energy_thresh, target_level).item())
