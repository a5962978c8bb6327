"""Phase timing of the forward FFT STFT kernel (debug build with -DFFT_STAMPS): cycles workgroup (1, 0) spends per phase."""
import ctypes
import math
import os
import sys

import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1] if len(sys.argv) > 1 else "libfft_dbg.so"))
lib.noise_last_error.restype = ctypes.c_char_p
P, I = ctypes.c_void_p, ctypes.c_int
lib.noise_stft_fwd_fft.argtypes = [P, P, P, P, P, P, P, I, I, I, I, I, P]
lib.noise_debug_fft_stamps.argtypes = [P, I]
dev = torch.device("cuda:0")


def run(B, L, n_fft, hop, fast):
    n2 = n_fft // 2
    wave = torch.randn(B, L, device=dev) * 0.1
    win = torch.hann_window(n_fft, periodic=True, device=dev)
    a1 = -2.0 * math.pi * torch.arange(n2, dtype=torch.float64) / n2
    a2 = -2.0 * math.pi * torch.arange(n2 + 1, dtype=torch.float64) / n_fft
    tw1 = torch.stack([torch.cos(a1), torch.sin(a1)], 1).float().to(dev)
    tw2 = torch.stack([torch.cos(a2), torch.sin(a2)], 1).float().to(dev)
    T, F = L // hop + 1, n2 + 1
    lm = torch.empty(B, F, T, device=dev)
    ph = torch.empty(B, F, T, device=dev)
    am = torch.empty(B, device=dev)
    pp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call():
        rc = lib.noise_stft_fwd_fft(pp(wave), pp(win), pp(tw1), pp(tw2), pp(lm), pp(ph), pp(am), B, L, n_fft, hop, fast, st)
        assert rc == 0, lib.noise_last_error()
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    lib.noise_debug_fft_stamps(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib.noise_debug_fft_stamps(out, 0)
    ms = e0.elapsed_time(e1) / n
    nb = 4 * B * L + 8 * B * F * T
    print(f"n_fft={n_fft} hop={hop} B={B} L={L} fast={fast}: {ms:.3f} ms  {nb / ms / 1e6:.0f} GB/s")
    names = ["tables+staging issue", "barrier (loads land)", "pass A", "barrier", "pass B", "barrier", "split+epilogue"]
    tot = sum(out[i] for i in range(7))
    for i, nm in enumerate(names):
        print(f"   {nm:22s} {out[i] / n:10.0f} ticks  {100.0 * out[i] / max(tot, 1):5.1f} %")


if __name__ == "__main__":
    run(256, 160000, 320, 160, 1)
    run(256, 160000, 512, 128, 1)
