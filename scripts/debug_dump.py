"""Dump forward/backward tensors of the golden train step: python scripts/debug_dump.py out.pt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_model import build, _keep
from noise_amd import engine
dev = torch.device("cuda:0")
g = np.load("tests/golden/g3_train_step.npz")
m = build(320, 160, dev, "tests/golden").train()
opt = m.configure_optimizers()
noisy, clean = torch.from_numpy(g["noisy"]).to(dev), torch.from_numpy(g["clean"]).to(dev)
m.attn_keep_mask = _keep(g, "attn_keep1", dev)
saved = {}
orig = engine.DenoiserPlan.backward
def bw(self, d_enh, d_vad, grad_of):
    saved["d_enh"] = d_enh.clone().cpu(); saved["d_vad"] = d_vad.clone().cpu(); saved["enh"] = self.enh.clone().cpu()
    for r in self.recs:
        if r.y is not None:
            saved["y/" + r.name] = r.y.clone().cpu(); saved["coef/" + r.name] = r.coef.clone().cpu()
    orig(self, d_enh, d_vad, grad_of)
    for r in self.recs:
        if getattr(r, "g_in", None) is not None:
            saved["g_in/" + r.name] = r.g_in.clone().cpu()
        if r.y is not None:
            saved["coef3/" + r.name] = r.coef3.clone().cpu()
engine.DenoiserPlan.backward = bw
m.train_step_fused(noisy, clean, opt)
torch.save(saved, sys.argv[1])
