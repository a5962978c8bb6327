"""Accuracy of the FFT STFT / iSTFT kernels (library vs hardware-rate transcendental functions) against float64, and their
rates at the inference and training sizes.

    python scripts/stft_accuracy.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from noise_amd import STFT

dev = torch.device("cuda:0")


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n_fft, hop in ((320, 160), (512, 128)):
    st = STFT(1300, n_fft, hop, n_fft, "hann").to(dev)
    g = torch.Generator().manual_seed(n_fft)
    w = (torch.randn(4, 1, 1, 100 * hop, generator=g) * 0.1).clamp(-1, 1)
    w[1] *= 0.01
    w[2] = w[2] * torch.linspace(0, 1, 100 * hop) ** 4
    # float64 reference of the spectrum: torch.stft with the reference's framing
    X = torch.stft(w.view(4, -1).double(), n_fft, hop, n_fft, torch.hann_window(n_fft, periodic=True, dtype=torch.float64),
                   center=True, pad_mode="reflect", return_complex=True)
    for fast in (False, True):
        lm, ph = st.forward(w.to(dev), fast_math=fast)
        Xg = torch.exp(lm[:, 0].cpu().double()) * torch.exp(1j * ph.cpu().double())
        live = X.abs() > 1e-3 * (1 + 1e-6)
        err = ((Xg - X).abs()[live]).max().item() / X.abs().max().item()
        lerr = (lm[:, 0].cpu().double() - X.abs().clamp_min(1e-3).log())[live].abs().max().item()
        perr = torch.angle(torch.exp(1j * (ph.cpu().double() - torch.angle(X))))[live].abs().max().item()
        # inverse from exact float64-derived inputs
        lm64, ph64 = X.abs().clamp_min(1e-3).log().float(), torch.angle(X).float()
        rec = st._inverse_raw(lm64.view(4, 1, *lm64.shape[1:]).to(dev), ph64.to(dev), None, fast_math=fast).cpu().view(4, -1)
        ref = torch.istft(torch.polar(lm64.double().exp(), ph64.double()), n_fft, hop, n_fft,
                          torch.hann_window(n_fft, periodic=True, dtype=torch.float64), center=True, length=rec.shape[1])
        ierr = (rec.double() - ref.clamp(-1, 1)).abs()[:, n_fft:-n_fft].max().item()
        print(f"n_fft {n_fft} hop {hop} fast={int(fast)}: spectrum err/max {err:.2e}  logmag abs {lerr:.2e}  phase abs {perr:.2e}  "
              f"istft abs {ierr:.2e}")
    for B, L, tag in ((256, 160000, "256x10s"), (1280, 8000, "1280x0.5s")):
        x = torch.randn(B, 1, 1, L, device=dev) * 0.1
        for fast in (False, True):
            lm, ph = st.forward(x, fast_math=fast)
            nb = 4 * B * L + 8 * lm.numel()
            tf = timed(lambda: st.forward(x, fast_math=fast))
            tn = timed(lambda: st.forward(x, fast_math=fast, want_phase=False))
            ti = timed(lambda: st._inverse_raw(lm, ph, None, fast_math=fast))
            print(f"  {tag} fast={int(fast)}: fwd {tf:.3f} ms {nb / 1e6 / tf:6.0f} GB/s | fwd no phase {tn:.3f} ms "
                  f"{(nb - 4 * lm.numel()) / 1e6 / tn:6.0f} GB/s | inv {ti:.3f} ms {nb / 1e6 / ti:6.0f} GB/s", flush=True)
        del x, lm, ph
