import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    x, y = a[k].double(), b[k].double()
    d = (x - y).abs().max().item()
    print(f"{k:40s} maxdiff {d:.3e} scale {y.abs().max().item():.3e} ndiff>1e-3*scale {(((x-y).abs() > 1e-3*y.abs().max()).sum().item())}")
ea, eb = a["enh"], b["enh"]
mx = torch.maximum(ea.abs().amax((1, 2), keepdim=True), eb.abs().amax((1, 2), keepdim=True))
sa, sb = ea.abs() >= mx, eb.abs() >= mx
flip = (sa != sb).nonzero()
print("saturated", sa.sum().item(), sb.sum().item(), "flips", flip.tolist())
for f in flip.tolist():
    print(f, ea[tuple(f)].item(), eb[tuple(f)].item(), mx[f[0]].item(), "d_enh", a["d_enh"][tuple(f)].item())
