import copy, sys
import numpy as np, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
from make_golden_ofa import choose, perturb
from noise_amd.ofa import OFASpectralNets
from oracle import ofa as oofa
seed = 7
rng = np.random.RandomState(100 + seed)
ks_list = [[3, 5, 7], [3, 7], [5], [3, 5]][seed % 4]
mults = [[0.5, 0.75, 1.0], [1.0], [0.5, 1.0]][seed % 3]
torch.manual_seed(seed)
net = OFASpectralNets(base_stage_width=[16, 8, 16, 24, 24, 32, 40, 48, 64], kernel_size_list=ks_list, expand_ratio_list=[2, 3], width_mult_list=mults)
perturb(net, seed); choose(net, ks=int(rng.choice(ks_list)), wid=int(rng.randint(len(mults))))
ref = copy.deepcopy(net); net = net.cuda()
B, H, W = int(rng.randint(1, 4)), int(rng.randint(20, 90)), int(rng.randint(20, 70))
x = torch.from_numpy(rng.randn(B, 1, H, W).astype(np.float32))
net.train(); ref.train()
oofa.forward(ref, x); net(x.cuda())
print("padding", net.H_padding, net.W_padding, ref.H_padding, ref.W_padding)
bad = 0
for (k, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
    if "running" in k:
        a = a.cpu().numpy(); b = b.numpy()
        e = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
        if e > 1e-3 and bad < 12:
            i = int(np.abs(a - b).argmax())
            print(f"{k:60s} {e:.2e}  at {i}: {a[i]:.6g} vs {b[i]:.6g}")
            bad += 1
