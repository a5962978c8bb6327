import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from noise_amd import STFT
from noise_amd.data import synth_batch
dev = torch.device("cuda:0")
for n_fft, hop in ((320, 160), (512, 128)):
    st = STFT(1300, n_fft, hop, n_fft, "hann").to(dev)
    for B, L, seed in ((1280, 8000, 1), (64, 160000, 2), (7, 12345, 3)):
        x = synth_batch(B, L, dev, seed=seed)[0]
        x[0] = 0                      # a silent clip: every bin on the floor
        x[1] = x[1] * 1e-3            # near the floor everywhere
        x[2, ..., : L // 2] = 0
        ld, pd, ad = st.forward(x, return_absmax=True, reference_order=True)
        lf, pf, af = st.forward(x, return_absmax=True, reference_order="normaliser")
        lp, pp, ap = st.forward(x, return_absmax=True)
        nd = (ad != af).sum().item()
        print(n_fft, hop, B, L, "refined != dense:", nd, "of", B, "| plain fft != dense:", (ad != ap).sum().item(),
              "| max rel diff refined", ((ad - af).abs() / ad).max().item(), "plain", ((ad - ap).abs() / ad).max().item(),
              "| logmag max diff", (ld - lf).abs().max().item())
        assert torch.equal(lf, lp) and torch.equal(pf, pp)
    import time
    x = synth_batch(1280, 8000, dev, seed=5)[0]
    for mode in (True, "normaliser", False):
        for _ in range(3): st.forward(x, return_absmax=True, reference_order=mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): st.forward(x, return_absmax=True, reference_order=mode)
        torch.cuda.synchronize(); print("  mode", mode, (time.perf_counter() - t0) / 20 * 1e3, "ms")
