import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import test_gpu_model as T
dev = torch.device("cuda:0")
gd = os.path.join(os.path.dirname(T.__file__), "golden")
res = {}
for combo in ("dense", "fft_ref"):
    g, m, tame = T._load_train_fixture(gd, "g3t_train_step.npz", dev)
    st = m.stft
    orig = st.forward
    def fwd(w, return_absmax=False, reference_order=False, want_phase=True, fast_math=None, combo=combo):
        if not reference_order:
            return orig(w, return_absmax=return_absmax, want_phase=want_phase, fast_math=fast_math)
        return orig(w, return_absmax=True, reference_order="dense" if combo == "dense" else True)
    st.forward = fwd
    opt = m.configure_optimizers()
    noisy, clean = torch.from_numpy(g["noisy"]).to(dev), torch.from_numpy(g["clean"]).to(dev)
    m.attn_keep_mask = T._keep(g, "attn_keep1", dev)
    m.train_step_fused(noisy, clean, opt)
    B = noisy.numel() // 8000
    plan = m.model.plan(B, 161, 51, dev)
    bufs = list(m._step_bufs.values())[0]
    res[combo] = dict(enh=plan.enh.clone(), d_enh=bufs["d_enh"].clone(), vad=plan.vad.clone(), d_wave=bufs["d_wave"].clone(),
                      x3=plan.x3.clone(), y0=plan.enc[0][0].y.clone(), lm=plan.logmag.clone(),
                      gl=plan.dec[-1][0].g_in.clone(), g3=plan.dec[0][0].g_in.clone())
a, b = res["dense"], res["fft_ref"]
for k in a:
    d = (a[k] - b[k]).abs()
    print(f"{k:8s} max|a| {a[k].abs().max().item():.3e}  max diff {d.max().item():.3e}  mean diff {d.mean().item():.3e}  "
          f"n(diff > 1e-3 max) {(d > 1e-3 * a[k].abs().max()).sum().item()} of {d.numel()}")
d = (a["d_enh"] - b["d_enh"]).abs()
idx = torch.nonzero(d > 1e-3 * a["d_enh"].abs().max())[:12]
for i in idx:
    i = tuple(i.tolist())
    print(i, "d_enh", a["d_enh"][i].item(), b["d_enh"][i].item(), "enh", a["enh"][i].item(), b["enh"][i].item(), "lm", a["lm"][i].item(), b["lm"][i].item())
