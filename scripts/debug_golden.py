import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_model import build, _keep
dev = torch.device("cuda:0")
g = np.load("tests/golden/g3_train_step.npz")
m = build(320, 160, dev, "tests/golden").train()
opt = m.configure_optimizers()
noisy, clean = torch.from_numpy(g["noisy"]).to(dev), torch.from_numpy(g["clean"]).to(dev)
print("shape", noisy.shape)
m.attn_keep_mask = _keep(g, "attn_keep1", dev)
out4 = m.train_step_fused(noisy, clean, opt).cpu().numpy()
print(out4, [g[f"log/train_{k}"] for k in ("loss", "vad_loss", "waveform_loss", "spectral_loss")])
rows = []
for k, p in m.named_parameters():
    ref = g["grad/" + k]
    sc = max(np.abs(ref).max(), 1e-4)
    rows.append((np.abs(p.grad.cpu().numpy() - ref).max() / sc, np.abs(ref).max(), k))
for e, s, k in rows:
    if "decoder.3" in k or "decoder.4" in k or "decoder.2" in k:
        print(f"{k:60s} err {e:.2e} scale {s:.2e}")
