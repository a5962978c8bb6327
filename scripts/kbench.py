"""Micro-benchmark of single entry points at the training step's shapes (for rocprofv3 --pmc runs).

    python scripts/kbench.py pw_bwd 72 24 891480      # Cout Cin npix
    python scripts/kbench.py dw_bwd 48 81 33 2 0      # C Hi Wi stride transposed
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_amd import ops

dev = torch.device("cuda:0")
which = sys.argv[1]
args = [int(a) for a in sys.argv[2:]]
reps = 5


def coef(C, n):
    return (torch.rand(n * C, device=dev) + 0.5)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if which == "pw_bwd":
    N, K, npix = args
    dz, y = torch.randn(npix, N, device=dev), torch.randn(npix, N, device=dev)
    a = torch.randn(npix, K, device=dev)
    w = torch.randn(N, K, device=dev)
    g_in, dw = torch.empty(npix, K, device=dev), torch.empty(N, K, device=dev)
    scratch = torch.empty(ops.pw_bwd_max_rows(N, K) * N * K, device=dev)
    stats = torch.empty(1024 * 2 * K, device=dev)
    ms = timed(lambda: ops.pw_bwd((dz, y, coef(N, 6)), (a, coef(K, 5), True), None, w, g_in, dw, scratch, npix, N, K,
                                  None, (coef(K, 5), True), stats))
    nbytes = 4 * npix * (2 * N + 2 * K)
elif which == "dw_bwd":
    C, Hi, Wi, s, tr = args
    B = 1280
    Ho, Wo = ops.conv_out(Hi, s, bool(tr)), ops.conv_out(Wi, s, bool(tr))
    dz, y = torch.randn(B, Ho, Wo, C, device=dev), torch.randn(B, Ho, Wo, C, device=dev)
    a = torch.randn(B, Hi, Wi, C, device=dev)
    w = torch.randn(C, 1, 3, 3, device=dev)
    g_in, dw = torch.empty(B, Hi, Wi, C, device=dev), torch.empty(C, 1, 3, 3, device=dev)
    scratch = torch.empty(ops.CONV_MAX_BLOCKS * 12 * C, device=dev)
    ms = timed(lambda: ops.dw3x3_bwd((dz, y, coef(C, 6)), (a, coef(C, 5), True), None, w, g_in, dw, scratch, bool(tr), B, Hi,
                                     Wi, C, (s, s), None, (coef(C, 5), True)))
    nbytes = 4 * B * C * (2 * Ho * Wo + 2 * Hi * Wi)
elif which == "stft":
    n_fft, hop, B, L = args
    from noise_amd import STFT
    st = STFT(L // hop + 1, n_fft, hop, n_fft, "hann").to(dev)
    x = torch.randn(B, 1, 1, L, device=dev) * 0.1
    lm, ph = st.forward(x)
    g = torch.randn(B, 1, 1, hop * (L // hop), device=dev)
    out = st._inverse_raw(lm, ph, None)
    ms = timed(lambda: st.forward(x, return_absmax=True))
    ms_i = timed(lambda: st._inverse_raw(lm, ph, None))
    ms_b = timed(lambda: st._inverse_bwd(g, out, lm, ph))
    nbytes = 4 * B * L + 8 * lm.numel()
    print(f"istft fwd {ms_i:.3f} ms, istft bwd {ms_b:.3f} ms")
else:
    raise SystemExit("unknown kernel")
print(f"{which} {args}: {ms:.3f} ms  {nbytes / 1e6 / ms:.1f} GB/s")
