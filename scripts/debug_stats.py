import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from noise_amd import VADSpectralV2
from noise_amd.data import synth_batch
from oracle import spectral as osp
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
keep = torch.rand(B, 6, 55, 55, generator=torch.Generator().manual_seed(1)) > 0.1
full = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
ns = {}
osp.shared_step(noisy.double(), clean.double(), full, tuple(b.double() for b in osp.stft_buffers(1200, 320, 160)), 320, 160, training=True, new_stats=ns, attn_keep=keep)
m.attn_keep_mask = keep.to(torch.uint8).to(dev).contiguous()
with torch.no_grad():
    m.training_step((noisy.to(dev), clean.to(dev), names), 0)
bufs = dict(m.named_buffers())
rows = []
for k, v in ns.items():
    got = bufs[k].cpu().double()
    if "running_var" in k:
        bv = (v - 0.9) / 0.1; bg = (got - 0.9) / 0.1
        mean = ns[k.replace("running_var", "running_mean")] * 10
        rel = ((bg - bv).abs() / bv.abs().clamp_min(1e-12))
        i = int(rel.argmax())
        rows.append((float(rel.max()), k, float(bv[i]), float(mean[i])))
for a, k, var, mean in sorted(rows, reverse=True)[:10]:
    print(f"{k:60s} batch-var rel err {a:.2e}  (var {var:.3e}, mean {mean:.3e})")
print("---- mask comparison")
rec_out = {}
orig_bn = osp._bn
def bn_hook(x, sd_, prefix, training, momentum=0.1, new_stats=None):
    y = orig_bn(x, sd_, prefix, training, momentum, new_stats)
    rec_out[prefix] = y.detach()
    return y
osp._bn = bn_hook
osp.shared_step(noisy.double(), clean.double(), full, tuple(b.double() for b in osp.stft_buffers(1200, 320, 160)), 320, 160, training=True, new_stats={}, attn_keep=keep)
plan = list(m.model._plans.values())[0]
name_map = {}
for r in plan.recs:
    if r.bn is None: continue
    for k, mod in m.named_modules():
        if mod is r.bn: name_map[r.name] = k
for r in plan.recs:
    if r.bn is None: continue
    ref = rec_out[name_map[r.name]].permute(0, 2, 3, 1)          # NHWC f64 BN output
    C = r.C
    v = (r.y.double() * r.coef[:C].double() + r.coef[C:2*C].double()).cpu()
    err = (v - ref).abs().max().item()
    if r.relu6:
        mg = (v > 0) & (v < 6); mr = (ref > 0) & (ref < 6)
        nm = int((mg != mr).sum())
        near = int(((ref.abs() < 1e-5) | ((ref - 6).abs() < 1e-5)).sum())
    else:
        nm = near = 0
    print(f"{r.name:16s} C={C:4d} max|v-ref| {err:.2e}  mask mismatches {nm:6d}  near-boundary(1e-5) {near}")
print("---- d_enh comparison")
osp._bn = orig_bn
store = {}
orig_sf = osp.spectral_forward
def sf_hook(logmag, sd_, **kw):
    enh, vad = orig_sf(logmag, sd_, **kw)
    enh.retain_grad(); vad.retain_grad()
    store["enh"], store["vad"] = enh, vad
    return enh, vad
osp.spectral_forward = sf_hook
params = {k: v.clone().requires_grad_(True) for k, v in full.items() if v.is_floating_point() and not ("running" in k or k.startswith("stft."))}
f2 = dict(full); f2.update(params)
loss, _ = osp.shared_step(noisy.double(), clean.double(), f2, tuple(b.double() for b in osp.stft_buffers(1200, 320, 160)), 320, 160, training=True, new_stats={}, attn_keep=keep)
loss.backward()
m.load_state_dict(sd)
out = m.training_step((noisy.to(dev), clean.to(dev), names), 0); out.backward()
b = list(m._step_bufs.values())[0]
ref = store["enh"].grad[:, 0]
got = b["d_enh"].cpu().double()
d = (got - ref).abs()
print("d_enh max abs err", d.max().item(), "ref max", ref.abs().max().item())
for i in range(4):
    print(" clip", i, "max err", d[i].max().item(), "n(err>1e-9)", int((d[i] > 1e-9).sum()), "ref absmax", ref[i].abs().max().item())
print("d_vad", b["d_vad"].cpu().numpy(), store["vad"].grad.flatten().numpy())
print("---- dz per layer vs f64 oracle (backward order)")
zs = {}
def bn_hook2(x, sd_, prefix, training, momentum=0.1, new_stats=None):
    y = orig_bn(x, sd_, prefix, training, momentum, new_stats)
    y.retain_grad(); zs[prefix] = y
    return y
osp._bn = bn_hook2
osp.spectral_forward = orig_sf
params = {k: v.clone().requires_grad_(True) for k, v in full.items() if v.is_floating_point() and not ("running" in k or k.startswith("stft."))}
f2 = dict(full); f2.update(params)
loss, _ = osp.shared_step(noisy.double(), clean.double(), f2, tuple(b.double() for b in osp.stft_buffers(1200, 320, 160)), 320, 160, training=True, new_stats={}, attn_keep=keep)
loss.backward()
for r in reversed(plan.recs):
    if r.bn is None: continue
    ref = zs[name_map[r.name]].grad.permute(0, 2, 3, 1)
    got = r.g1.double().cpu()
    d = (got - ref).abs()
    per_clip = [f"{d[i].max().item() / ref.abs().max().item():.1e}" for i in range(4)]
    print(f"{r.name:16s} rel max err {d.max().item() / ref.abs().max().item():.2e}  per clip {per_clip}  n>1e-3: {int((d > 1e-3 * ref.abs().max()).sum())}")
