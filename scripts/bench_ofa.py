"""Once-for-all spectral U-Net forward (train/layers.py OFASpectralNets) on the generic kernels: per-entry-point table.

    python scripts/bench_ofa.py [batch] [ks]      # batch of 161 x 101 spectra (1 s of audio at 320/160 each)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_amd import _lib
from noise_amd.ofa import OFASpectralNets

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ks = int(sys.argv[2]) if len(sys.argv) > 2 else 7
torch.manual_seed(0)
net = OFASpectralNets().to(dev)
net.set_active(ks=ks)
x = torch.randn(B, 1, 161, 101, device=dev)
for mode in ("train", "eval"):
    net.train(mode == "train")
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        net(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    _lib.PROFILE = prof = []
    net(x)
    torch.cuda.synchronize()
    _lib.PROFILE = None
    agg = {}
    for name, nbytes, a, b in prof:
        r = agg.setdefault(name, [0.0, 0, 0])
        r[0] += a.elapsed_time(b)
        r[1] += nbytes
        r[2] += 1
    print(f"== {mode}: {B} x 161 x 101 spectra, kernel size {ks}: {ms:.3f} ms/forward, {len(prof)} launches")
    for name, (t, nb, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"   {name:24s} {t:8.3f} ms  {n:4d} calls  {nb / 1e6:9.1f} MB  {nb / 1e6 / max(t, 1e-9):8.1f} GB/s")
    if os.environ.get("OFA_CALLS"):
        for i, (name, nbytes, a, b) in enumerate(prof):
            t = a.elapsed_time(b)
            if t > 0.03:
                print(f"   #{i:3d} {name:22s} {t:7.3f} ms {nbytes / 1e6:8.1f} MB {nbytes / 1e6 / t:8.1f} GB/s")
    print(json.dumps({"ofa_forward": {"mode": mode, "batch": B, "ks": ks, "ms": round(ms, 3),
                                      "spectra_per_s": round(B / ms * 1e3, 1)}}))
