"""Large-batch sanity run of the waveform augmentation (diagnostic): 1024 pairs of 10 s, a 3 s impulse response (47
partitions), a unit impulse response (the chain must equal the chain without reverb) and a 1025-tap one."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from noise_amd import augment
rng = np.random.RandomState(0)
L = 160000
N, B = 32, 1024
cl = (rng.randn(N, L) * 0.05).astype(np.float32)
ny = (cl + rng.randn(N, L).astype(np.float32) * 0.02).astype(np.float32)
irs = [(rng.randn(48000) * np.exp(-np.arange(48000) / 9000.0)).astype(np.float32), np.array([1.0], dtype=np.float32),
       (rng.randn(1025) * 0.1).astype(np.float32)]
pools = augment.AugmentPools(ny, cl, [(rng.randn(L // 3) * 0.05).astype(np.float32)], irs)
t = augment.VADTrainAugment(pools, 8000, p=0.5, seed=0)
items = t.draw([i % N for i in range(B)])
a, c = t.apply(items)
torch.cuda.synchronize()
assert torch.isfinite(a).all() and a.abs().max() <= 1
# unit impulse response: the chain equals the no-reverb chain up to the RMS match (scale 1) and clip
it2 = items.copy(); it2["ir"] = np.where(it2["ir"] >= 0, 1, -1); it2 = t.assign_rows(it2)
a2, _ = t.apply(it2)
it3 = items.copy(); it3["ir"] = -1; it3 = t.assign_rows(it3)
a3, _ = t.apply(it3)
print("unit-ir vs none max diff", (a2 - a3).abs().max().item(), "rows with ir", int((items["ir"] >= 0).sum()))
assert (a2 - a3).abs().max().item() < 2e-6
print("ok", a.shape)
