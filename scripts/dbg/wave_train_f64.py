"""GPU training-step gradients against a float64 CPU evaluation of the same step, beside the fp32 CPU oracle's distance from
the same yardstick (diagnostic for ill-conditioned fixtures)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import test_gpu_wave as T
from oracle import wave_unet as owu
name = sys.argv[1] if len(sys.argv) > 1 else "yaml"
cfg = T.MG.CASES[name][0]
x1c, cleanc = torch.from_numpy(T.G9[f"{name}_x1"]), torch.from_numpy(T.G9[f"{name}_val_clean"])
res = {}
for dt in (torch.float32, torch.float64):
    m = T.build(name, 1.0).eval()
    owu.forward(m, owu.normalize(x1c)[0], False)
    m.train()
    nl = len(m.bottleneck_attention.layer_stack)
    res[dt] = owu.train_step(m, x1c, cleanc, T.g10_masks(name, nl), cfg.get("mr_stft_lambda", 0.5), [512, 1024, 2048],
                             [50, 120, 240], [240, 600, 1200], dtype=dt)[1]
m = T.build(name, 1.0).to(T.DEV).eval()
m(x1c.to(T.DEV)); m.train()
m.attn_keep_masks = T.g10_masks(name, nl)
loss = m.training_step((x1c.to(T.DEV), cleanc.to(T.DEV), None), 0); loss.backward()
for k, p in m.named_parameters():
    r64 = res[torch.float64][k].double()
    n = float(r64.norm())
    eg = float((p.grad.cpu().double() - r64).norm()) / max(n, 1e-30)
    ec = float((res[torch.float32][k].double() - r64).norm()) / max(n, 1e-30)
    print(f"gpu {eg:9.2e}  cpu32 {ec:9.2e}  |g64| {n:9.2e}  {k}")
