import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_model import build
from oracle import spectral as osp
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
m = build(320, 160, dev, "tests/golden").train()
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
gen = torch.Generator().manual_seed(100 + B)
clean = (torch.randn(B, 1, 1, 8000, generator=gen) * 0.05).clamp(-1, 1)
if B > 2:
    clean[1] = 0
noisy = (clean + 0.03 * torch.randn(B, 1, 1, 8000, generator=gen)).clamp(-1, 1)
keep = torch.rand(B, 6, 55, 55, generator=gen) > 0.1
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and not ("running" in k or k.startswith("stft."))}
full = dict(sd); full.update(params)
bufs = osp.stft_buffers(1200, 320, 160)
loss, logs = osp.shared_step(noisy, clean, full, bufs, 320, 160, training=True, new_stats={}, attn_keep=keep)
loss.backward()
m.attn_keep_mask = keep.to(torch.uint8).to(dev).contiguous()
out = m.training_step((noisy.to(dev), clean.to(dev), [""] * B), 0)
out.backward()
print("loss", out.item(), loss.item())
# double-precision oracle for reference noise level
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
p64 = {k: v.detach().double().clone().requires_grad_(True) for k, v in params.items()}
full64 = dict(sd64); full64.update(p64)
bufs64 = tuple(b.double() for b in bufs)
l64, _ = osp.shared_step(noisy.double(), clean.double(), full64, bufs64, 320, 160, training=True, new_stats={}, attn_keep=keep)
l64.backward()
print("loss64", l64.item())
rows = []
for k, p in m.named_parameters():
    r64 = p64[k].grad.numpy()
    sc = max(np.abs(r64).max(), 1e-4)
    e_gpu = np.abs(p.grad.cpu().numpy() - r64).max() / sc
    e_cpu = np.abs(params[k].grad.numpy() - r64).max() / sc
    rows.append((e_gpu, e_cpu, k))
for e_gpu, e_cpu, k in sorted(rows, reverse=True)[:15]:
    print(f"{k:70s} gpu-vs-f64 {e_gpu:.2e}   cpu32-vs-f64 {e_cpu:.2e}")
