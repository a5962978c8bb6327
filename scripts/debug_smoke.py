import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from noise_amd import VADSpectralV2
from noise_amd.data import synth_batch
from tests.test_gpu_model import _oracle_grads
dev = torch.device("cuda:0")
lc = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [3, 24, 1, [3, 3], [2, 2]], [3, 32, 1, [3, 3], [2, 2]], [3, 64, 1, [3, 3], [2, 2]]]
torch.manual_seed(0)
m = VADSpectralV2(n_fft=320, hop_length=160, win_length=320, window="hann", nhead=6, num_layers=1, dropout=0.0, layer_config=lc, pad_value=-1, max_frames=1200)
with torch.no_grad():
    m.model.decoder[-1][0].weight.mul_(0.02); m.model.decoder[-1][0].bias.zero_()
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
m = m.to(dev).train()
B = 4
noisy, clean, names = synth_batch(B, 8000, "cpu", seed=7)
print("clean rms per clip", clean.pow(2).mean(-1).sqrt().flatten())
keep = torch.rand(B, 6, 55, 55, generator=torch.Generator().manual_seed(1)) > 0.1
l32, g32 = _oracle_grads(sd, noisy, clean, keep, torch.float32)
l64, g64 = _oracle_grads(sd, noisy, clean, keep, torch.float64)
m.attn_keep_mask = keep.to(torch.uint8).to(dev).contiguous()
loss = m.training_step((noisy.to(dev), clean.to(dev), names), 0); loss.backward()
print(loss.item(), l32, l64)
rows = []
for k, p in m.named_parameters():
    r = g64[k]; sc = max(np.abs(r).max(), 1e-4)
    rows.append((np.abs(p.grad.cpu().double().numpy() - r).max() / sc, np.abs(g32[k] - r).max() / sc, k))
for a, b, k in sorted(rows, reverse=True)[:12]:
    print(f"{k:70s} gpu {a:.2e} cpu32 {b:.2e}")
print("---- no silent clip")
clean2 = clean.clone(); clean2[0] = clean[1].flip(-1)
noisy2 = noisy.clone(); noisy2[0] = (clean2[0] + 0.01 * torch.randn(1, 1, 8000)).clamp(-1, 1)
l32, g32 = _oracle_grads(sd, noisy2, clean2, keep, torch.float32)
l64, g64 = _oracle_grads(sd, noisy2, clean2, keep, torch.float64)
m.zero_grad()
m.load_state_dict(sd)
loss = m.training_step((noisy2.to(dev), clean2.to(dev), names), 0); loss.backward()
print(loss.item(), l32, l64)
rows = []
for k, p in m.named_parameters():
    r = g64[k]; sc = max(np.abs(r).max(), 1e-4)
    rows.append((np.abs(p.grad.cpu().double().numpy() - r).max() / sc, np.abs(g32[k] - r).max() / sc, k))
for a, b, k in sorted(rows, reverse=True)[:6]:
    print(f"{k:70s} gpu {a:.2e} cpu32 {b:.2e}")
# forward-only accuracy of batch statistics: compare BN running stats after the step
from oracle import spectral as osp
full = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
ns = {}
osp.shared_step(noisy2.double(), clean2.double(), full, tuple(b.double() for b in osp.stft_buffers(1200, 320, 160)), 320, 160, training=True, new_stats=ns, attn_keep=keep)
rows = []
for k, v in ns.items():
    got = dict(m.named_buffers())[k].cpu().double()
    rows.append((float(((got - v).abs() / (v.abs() + 1e-3)).max()), k))
for a, k in sorted(rows, reverse=True)[:6]:
    print(f"{k:70s} running-stat rel err {a:.2e}")
