"""(diagnostic, not collected by pytest; lives under tests/ because it checks against the CPU oracle)
Compare a fused training step with the folded STFT kernels against the general ones (same weights, same batch)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
dev = torch.device("cuda:0")
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
res = {}
for sym in (False, True):
    g, m, tame = T._load_train_fixture(gd, "g3t_train_step.npz", dev)
    m.stft.use_symmetry = sym
    noisy = torch.from_numpy(g["noisy"]).to(dev); clean = torch.from_numpy(g["clean"]).to(dev)
    keep = T._keep(g, "attn_keep1", dev)
    lm, ph, am = m.stft.forward(noisy.view(-1, 1, 1, 8000), return_absmax=True)
    opt = m.configure_optimizers()
    m.attn_keep_mask = keep
    out4 = m.train_step_fused(noisy, clean, opt, None)
    torch.cuda.synchronize()
    res[sym] = dict(lm=lm.cpu(), ph=ph.cpu(), am=am.cpu(), out4=out4.cpu(), grads=m._flat_grads.clone().cpu(),
                    enh=None)
a, b = res[False], res[True]
for k in ("lm", "ph", "am", "out4", "grads", "enh"):
    if a[k] is None: continue
    d = (a[k] - b[k]).abs()
    print(k, "max diff", float(d.max()), "ref max", float(a[k].abs().max()), "argmax", int(d.argmax()))
from oracle import spectral as osp
g = np.load(os.path.join(gd, "g3t_train_step.npz"))
noisy = torch.from_numpy(g["noisy"]).view(-1, 1, 1, 8000)
fwd, inv, wsum = osp.stft_buffers(1200, 320, 160)
lm64, ph64 = osp.stft_forward(noisy.double(), fwd.double(), 320, 160)
lm32, ph32 = osp.stft_forward(noisy, fwd, 320, 160)
for name, lm in (("general", a["lm"]), ("folded", b["lm"]), ("oracle32", lm32)):
    d = (lm.double() - lm64).abs()
    am = lm.abs().amax(dim=(1, 2, 3)).double(); am64 = lm64.abs().amax(dim=(1, 2, 3))
    X = torch.exp(lm.double()); X64 = torch.exp(lm64)
    print(f"{name:9s} max|dlm| {float(d.max()):.3e}  mean|dlm| {float(d.mean()):.3e}  max|dX| {float((X - X64).abs().max()):.3e} "
          f"mean|dX| {float((X - X64).abs().mean()):.3e}  absmax err {[f'{float(v):.1e}' for v in (am - am64).abs()]}")
g = np.load(os.path.join(gd, "g3t_train_step.npz"))
for sym in (False, True):
    gg, m, tame = T._load_train_fixture(gd, "g3t_train_step.npz", dev)
    m.stft.use_symmetry = sym
    m.attn_keep_mask = T._keep(gg, "attn_keep1", dev)
    opt = m.configure_optimizers()
    m.train_step_fused(torch.from_numpy(gg["noisy"]).to(dev), torch.from_numpy(gg["clean"]).to(dev), opt)
    num = den = 0.0; worst = 0.0
    for k, p in m.named_parameters():
        ref = gg["grad/" + k]; got = p.grad.cpu().numpy()
        num += float(((got - ref) ** 2).sum()); den += float((ref ** 2).sum())
        worst = max(worst, float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-4)))
    print("sym", sym, "global rel L2", (num / den) ** 0.5, "worst per-tensor", worst)
