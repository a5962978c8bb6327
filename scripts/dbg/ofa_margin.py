import copy, os, sys
import numpy as np, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
from make_golden_ofa import choose, perturb
from noise_amd.ofa import OFASpectralNets
from oracle import ofa as oofa
worst = 0
for seed in range(12):
    rng = np.random.RandomState(100 + seed)
    ks_list = [[3, 5, 7], [3, 7], [5], [3, 5]][seed % 4]
    mults = [[0.5, 0.75, 1.0], [1.0], [0.5, 1.0]][seed % 3]
    torch.manual_seed(seed)
    net = OFASpectralNets(base_stage_width=[16, 8, 16, 24, 24, 32, 40, 48, 64], kernel_size_list=ks_list, expand_ratio_list=[2, 3], width_mult_list=mults)
    perturb(net, seed); choose(net, ks=int(rng.choice(ks_list)), wid=int(rng.randint(len(mults))))
    ref = copy.deepcopy(net); net = net.cuda()
    B, H, W = int(rng.randint(1, 4)), int(rng.randint(20, 90)), int(rng.randint(20, 70))
    x = torch.from_numpy(rng.randn(B, 1, H, W).astype(np.float32))
    for mode in (True, True, False):
        net.train(mode); ref.train(mode)
        want = oofa.forward(ref, x).numpy(); got = net(x.cuda()).cpu().numpy()
        e = np.abs(got - want).max() / max(np.abs(want).max(), 1e-3)
        worst = max(worst, e)
        print(seed, mode, B, H, W, f"{e:.2e}")
print("worst", worst)
