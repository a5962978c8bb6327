"""Inference forward (STFT -> denoiser -> iSTFT) on 10 s clips: per-entry-point table and audio-seconds per second,
per-layer fp32 path vs block-fused bf16 path.   usage: bench_infer.py [A|B] [clips] [fp32|bf16|both]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from noise_amd import VADSpectralV2, _lib
from noise_amd.data import synth_batch

dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "A"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
which = sys.argv[3] if len(sys.argv) > 3 else "both"
n_fft, hop = (320, 160) if tag == "A" else (512, 128)
torch.manual_seed(0)
m = VADSpectralV2(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann", nhead=6, num_layers=1, dropout=0.0,
                  layer_config=[list(r) for r in bench.LAYER_CONFIG], pad_value=-1, max_frames=1300).to(dev).eval()
x, _, _ = synth_batch(B, 160000, dev, seed=7)
x = x.view(B, 1, 1, -1)
for mode in (("fp32", "bf16") if which == "both" else (which,)):
    m.inference_dtype = mode
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        _lib.PROFILE = prof = []
        m(x)
        torch.cuda.synchronize()
        _lib.PROFILE = None
    print(f"== {tag} {mode}: {B} x 10 s clips  {1000 * dt:.3f} ms/forward  {B * 10 / dt:,.0f} audio-s/s  ({1e6 * dt / B:.2f} us/clip)")
    tot = sum(e0.elapsed_time(e1) for _, _, e0, e1 in prof)
    for name, nb, e0, e1 in prof:
        ms = e0.elapsed_time(e1)
        if ms > 0.01 * tot:
            print(f"   {name:24s} {ms:8.3f} ms {100 * ms / tot:5.1f}%  {nb / 1e6:9.1f} MB  {nb / 1e6 / max(ms, 1e-9):8.1f} GB/s")
    print(f"   sum of entry points {tot:.3f} ms")
    m.model._plans.clear()
    torch.cuda.empty_cache()
