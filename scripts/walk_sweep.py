"""Sweep of band height / grid cap of the stencil walkers at the training step's shapes.

    python scripts/walk_sweep.py            # prints one line per (layer, setting)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_amd import ops

dev = torch.device("cuda:0")
B = 1280
MAXROWS = 1 << 16


def coef(C, n):
    return torch.rand(n * C, device=dev) + 0.5


def timed(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# (name, C, Hi, Wi, stride, transposed, two sources)
LAYERS = [("enc1.dw", 32, 81, 33, 1, 0, 0), ("enc2.dw", 48, 81, 33, 2, 0, 0), ("enc3.dw", 72, 41, 17, 2, 0, 0),
          ("dec3.dw", 16, 81, 33, 1, 1, 1), ("dec2.dw", 72, 41, 17, 2, 1, 0), ("dec1.dw", 96, 21, 9, 2, 1, 0)]
SETTINGS = [(0, 0), (8, 2048), (16, 2048), (8, 1024), (16, 1024), (8, 4096), (8, 65536), (12, 65536), (16, 65536),
            (24, 65536), (4, 65536)]
if os.environ.get("WALK_SWEEP") == "occupancy":      # fewer workgroups than the GPU can hold: how much does occupancy matter?
    SETTINGS = [(16, 256), (16, 512), (16, 768), (16, 1024), (16, 2048)]
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if l[0] in sys.argv[1:]]

for name, C, Hi, Wi, s, tr, two in LAYERS:
    Ho, Wo = ops.conv_out(Hi, s, bool(tr)), ops.conv_out(Wi, s, bool(tr))
    a = torch.randn(B, Hi, Wi, C, device=dev)
    b = (torch.randn(B, Hi, Wi, C, device=dev), None, False) if two else None
    w = torch.randn(C, 1, 3, 3, device=dev)
    out = torch.empty(B, Ho, Wo, C, device=dev)
    dz = torch.randn(B, Ho, Wo, C, device=dev)
    g_in = torch.empty(B, Hi, Wi, C, device=dev)
    go = torch.randn(B, Hi, Wi, C, device=dev)
    slab = torch.empty(MAXROWS * 12 * C, device=dev)
    ca, c3, ce = coef(C, 5), coef(C, 6), coef(C, 5)
    virt = tr and s == 2
    fb = 4 * B * C * (Hi * Wi * (2 if two else 1) + (0 if virt else Ho * Wo))
    bb = 4 * C * B * (Hi * Wi * ((2 if two else 1) + 2) + Ho * Wo * (1 if virt else 2))
    for bh, cap in SETTINGS:
        if bh:
            os.environ["NOISE_WALK_BH"] = str(bh)
            os.environ["NOISE_WALK_CAP"] = str(cap)
        else:
            os.environ.pop("NOISE_WALK_BH", None)
            os.environ.pop("NOISE_WALK_CAP", None)
        rows = [0, 0]

        def f():
            rows[0] = ops.dw3x3_fwd((a, ca, True), b, w, None if virt else out, slab, bool(tr), B, Hi, Wi, C, (s, s))

        def g():
            rows[1] = ops.dw3x3_bwd((dz, out, c3), (a, ca, True), b, w, g_in, None, slab, bool(tr), B, Hi, Wi, C, (s, s), go,
                                    (ce, True))

        tf, tb = timed(f), timed(g)
        print(f"{name:8s} C={C:3d} bh={bh:3d} cap={cap:6d}  fwd {tf:.3f} ms {fb / 1e6 / tf:7.0f} GB/s rows {rows[0]:6d} | "
              f"bwd {tb:.3f} ms {bb / 1e6 / tb:7.0f} GB/s rows {rows[1]:6d}", flush=True)
    del a, b, out, dz, g_in, go, slab
