"""Timing of the waveform models' loss kernels at the reference's training batch (1280 segments of 8000 samples)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from noise_amd import losses

CFG = ([512, 1024, 2048], [50, 120, 240], [240, 600, 1200])


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    B, L = 1280, 8000
    g = torch.Generator(device=dev).manual_seed(0)
    y = torch.randn(B, L, device=dev, generator=g) * 0.05
    x = y + 0.01 * torch.randn(B, L, device=dev, generator=g)
    ms_all = timed(lambda: losses.mrstft_loss_and_grad(x, y, *CFG))
    print(f"multi-resolution STFT loss + gradient, {B} x {L}: {ms_all:.3f} ms  ({B * L / 16000 / ms_all * 1e3:,.0f} audio-s/s)")
    for n, h, w in zip(*CFG):
        ms = timed(lambda: losses.mrstft_loss_and_grad(x, y, [n], [h], [w]))
        T = L // h + 1
        alg = 4 * (3 * B * L) + 8 * B * T * w          # x, y read, dx written; frame gradients written and read
        print(f"   n_fft {n:5d} hop {h:4d} win {w:5d}: {ms:.3f} ms  algorithmic {alg / 1e6:.0f} MB  {alg / ms / 1e6:.0f} GB/s")
        ms = timed(lambda: losses.mrstft_loss_and_grad(x, y, [n], [h], [w], want_grad=False))
        print(f"      loss only: {ms:.3f} ms")
    ms = timed(lambda: losses.smooth_l1_and_grad(x, y, 0.5))
    print(f"smooth-L1 + gradient: {ms:.3f} ms  {12 * B * L / ms / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
