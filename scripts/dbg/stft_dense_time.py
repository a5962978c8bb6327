"""Time the dense (reference-order) STFT forward, the network input of the training step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from noise_amd.stft import STFT

for B, L, n_fft, hop in ((1280, 8000, 320, 160), (1024, 8000, 320, 160), (256, 160000, 320, 160), (1280, 8000, 512, 128)):
    torch.manual_seed(0)
    st = STFT(max_frames=L // hop + 1, n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann").cuda()
    x = torch.randn(B, L, device="cuda") * 0.1
    for _ in range(3):
        st.forward(x, return_absmax=True, reference_order=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        st.forward(x, return_absmax=True, reference_order=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} L={L} n_fft={n_fft} hop={hop}: {e0.elapsed_time(e1) / 20:.4f} ms", flush=True)
