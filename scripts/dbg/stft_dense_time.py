"""Time the dense STFT kernels: the reference-order forward (network input of the training step), and the folded and
general forward / inverse that serve transforms the FFT kernels do not (use_fft off stands in for such a basis)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from noise_amd.stft import STFT


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


for B, L, n_fft, hop in ((1280, 8000, 320, 160), (256, 160000, 320, 160), (1280, 8000, 512, 128), (256, 160000, 1024, 256)):
    torch.manual_seed(0)
    x = torch.randn(B, L, device="cuda") * 0.1
    line = f"B={B} L={L} n_fft={n_fft} hop={hop}:"
    for sym in (True, False):
        st = STFT(max_frames=L // hop + 1, n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann").cuda()
        st.use_fft = False
        st.use_symmetry = sym
        if not sym:
            line += f"  reference-order fwd {timed(lambda: st.forward(x, return_absmax=True, reference_order=True)):.4f}"
        lm, ph = st.forward(x)
        line += f"  {'folded' if sym else 'general'} fwd {timed(lambda: st.forward(x)):.4f} inv {timed(lambda: st.inverse(lm, ph)):.4f}"
    print(line + " ms", flush=True)
