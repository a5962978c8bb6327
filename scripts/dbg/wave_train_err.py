"""Per-tensor gradient error of the generic training step against fixture G10 (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import test_gpu_wave as T
name = sys.argv[1] if len(sys.argv) > 1 else "narrow"
m = T.build(name, 1.0).to(T.DEV).eval()
x1, clean = torch.from_numpy(T.G9[f"{name}_x1"]).to(T.DEV), torch.from_numpy(T.G9[f"{name}_val_clean"]).to(T.DEV)
m(x1); m.train()
m.attn_keep_masks = T.g10_masks(name, len(m.bottleneck_attention.layer_stack))
loss = m.training_step((x1, clean, None), 0); loss.backward()
print("losses", float(loss), T.G10[f"{name}_losses"])
names, stats = list(T.G10[f"{name}_grad_names"]), T.G10[f"{name}_grad_stats"]
params = dict(m.named_parameters())
for k, (s, a, nrm) in zip(names, stats):
    g = params[k].grad
    key = f"{name}_grad/{k}"
    if key in T.G10.files:
        ref = torch.from_numpy(T.G10[key]).double()
        err = float((g.cpu().double() - ref).norm()) / max(nrm, 1e-12)
    else:
        err = abs(float(g.double().norm()) - nrm) / max(nrm, 1e-12)
    print(f"{err:10.3e}  |g| {float(g.double().norm()):10.3e} ref {nrm:10.3e}  {k} {tuple(g.shape)}")
