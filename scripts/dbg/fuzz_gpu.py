"""Randomised parity sweep on the GPU (diagnostic, not part of the test-suite): convolution and augmentation chain at
random lengths and parameters, STFT / iSTFT and the eval forward at random clip lengths, each against the CPU oracle."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from noise_amd import augment, VADSpectralV2  # noqa: E402
from oracle import augment as oa  # noqa: E402
from oracle import spectral as osp  # noqa: E402

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
t0 = time.time()
# 1. convolution
for it in range(40):
    L = int(rng.randint(1, 30000))
    n_ir = int(rng.randint(1, 14000))
    R = int(rng.randint(1, 5))
    x = rng.randn(R, L).astype(np.float32)
    irs = [(rng.randn(n_ir) * np.exp(-np.arange(n_ir) / (0.2 * n_ir + 1))).astype(np.float32),
           rng.randn(int(rng.randint(1, 3000))).astype(np.float32)]
    pool = augment.ImpulseResponsePool(irs)
    idx = rng.randint(0, 2, R)
    y = augment.fir_convolve(torch.from_numpy(x).cuda(), torch.from_numpy(idx.astype(np.int32)).cuda(), pool).cpu().numpy()
    for r in range(R):
        want = np.convolve(x[r].astype(np.float64), irs[idx[r]].astype(np.float64))[:L]
        err = np.abs(y[r] - want).max() / max(np.abs(want).max(), 1e-9)
        if not err < 3e-5:
            bad += 1
            print("CONV MISMATCH", L, n_ir, R, r, err)
print("conv done", time.time() - t0, flush=True)
# 2. augmentation chain
for it in range(12):
    N = 4
    L = int(rng.randint(9000, 50000))
    length = int(rng.randint(100, min(L, 9000)))
    clean = (rng.randn(N, L) * 0.1).astype(np.float32)
    noises = [(rng.randn(int(rng.randint(50, 2 * L))) * 0.1).astype(np.float32) for _ in range(3)]
    irs = [(rng.randn(int(rng.randint(1, 9000))) * 0.3).astype(np.float32) for _ in range(3)]
    for h in irs:
        h[0] = 1.0
    pools = augment.AugmentPools(clean, noises, irs)
    t = augment.VADTrainAugment(pools, length, p=0.5, seed=int(rng.randint(1 << 20)))
    items = t.draw(list(rng.randint(0, N, 9)), epoch=it)
    gn, gc = t.apply(items)
    pool_np = np.concatenate(noises)
    for n in range(len(items)):
        wn, wc = oa.vad_pair(clean[items[n]["src"]], items[n], pool_np, irs, length)
        e1 = np.abs(gn[n, 0, 0].cpu().numpy().astype(np.float64) - wn).max()
        e2 = np.abs(gc[n, 0, 0].cpu().numpy() - wc).max()
        if not (e1 <= 1e-5 and e2 == 0):
            bad += 1
            print("AUG MISMATCH", L, length, items[n], e1, e2)
print("aug done", time.time() - t0, flush=True)
# 3. eval forward at random lengths on one model (first padding kept, skips cropped), fp32
lc = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [3, 24, 1, [3, 3], [2, 2]], [3, 32, 1, [3, 3], [2, 2]],
      [3, 64, 1, [3, 3], [2, 2]]]
for n_fft, hop in ((320, 160), (512, 128)):
    torch.manual_seed(int(rng.randint(1 << 20)))
    m = VADSpectralV2(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann", nhead=6, num_layers=1, dropout=0.0,
                      layer_config=[list(r) for r in lc], pad_value=-1, max_frames=400)
    with torch.no_grad():
        m.model.decoder[-1][0].weight.mul_(0.05)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    bufs = osp.stft_buffers(400, n_fft, hop)
    m = m.cuda().eval()
    pad = None
    for it in range(5):
        L = int(rng.randint(20, 250)) * hop
        B = int(rng.randint(1, 4))
        x = (torch.randn(B, 1, 1, L) * 0.1).clamp(-1, 1)
        try:
            with torch.no_grad():
                y = m(x.cuda()).cpu()
        except (NotImplementedError, RuntimeError) as e:
            print("forward refused", n_fft, L, str(e)[:100])
            continue
        if pad is None:
            pad = (m.model.H_padding, m.model.W_padding)
        lm, ph = osp.stft_forward(x, bufs[0], n_fft, hop)
        enh_ref, vad_ref = osp.spectral_forward(lm, sd, padding=pad)
        with torch.no_grad():
            glm, gph = m.stft.forward(x.cuda())
            enh, vad = m._forward(glm)
        e_enh = (enh.cpu() - enh_ref).abs().max().item() / enh_ref.abs().max().item()
        e_vad = (vad.cpu() - vad_ref).abs().max().item()
        want = osp.model_forward(x, sd, bufs, n_fft, hop, padding=pad)
        gated = vad_ref.abs().min().item() > 1e-3
        err = (y - want).abs().max().item() if gated else 0.0
        print(f"  forward n_fft={n_fft} L={L} B={B} pad={pad}: enh {e_enh:.2e} vad {e_vad:.2e} wave {err:.2e}", flush=True)
        if not (e_enh < 2e-4 and e_vad < 2e-4 and err < 2e-4):
            bad += 1
            print("FORWARD MISMATCH", n_fft, L, B, e_enh, e_vad, err)
print("forward done", time.time() - t0, "mismatches:", bad, flush=True)
sys.exit(1 if bad else 0)
