"""Phase timing of the fused tail kernel (library built with -DFUSED_STAMPS, loaded through NOISE_HIP_LIB)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import bench
from noise_amd import VADSpectralV2, _lib
from noise_amd.data import synth_batch

dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "A"
n_fft, hop = (320, 160) if tag == "A" else (512, 128)
torch.manual_seed(0)
m = VADSpectralV2(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann", nhead=6, num_layers=1, dropout=0.0,
                  layer_config=[list(r) for r in bench.LAYER_CONFIG], pad_value=-1, max_frames=1300).to(dev).eval()
m.inference_dtype = "bf16"
x, _, _ = synth_batch(256, 160000, dev, seed=7)
x = x.view(256, 1, 1, -1)
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    _lib.lib.noise_debug_fused_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    _lib.lib.noise_debug_fused_stamps(None, 1)
    n = 5
    for _ in range(n):
        m(x)
    torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 32)()
_lib.lib.noise_debug_fused_stamps(out, 0)
names = ["prologue", "stage-in + barrier", "prefetch issue", "depthwise + barrier", "project + barrier", "tap MFMA + barrier",
         "output gather"]
print("tail kernel")
tot = sum(out[i] for i in range(7))
for i, nm in enumerate(names):
    print(f"   {nm:22s} {out[i] / n:10.0f} ticks  {100.0 * out[i] / max(tot, 1):5.1f} %")
print("head kernel")
hn = {8: "stage-in + barrier", 9: "prefetch issue", 10: "first conv + barrier", 11: "depthwise + barrier", 13: "project (+ loop top)"}
tot = sum(out[i] for i in hn)
for i, nm in hn.items():
    print(f"   {nm:22s} {out[i] / n:10.0f} ticks  {100.0 * out[i] / max(tot, 1):5.1f} %")
for base, title in ((16, "decoder block 24 -> 72 -> 16 (transposed)"), (24, "encoder block 16 -> 48 -> 24")):
    print(title)
    nm = {0: "stage-in + barrier", 1: "prefetch issue", 2: "expand + barrier", 3: "depthwise + barrier", 5: "project + stores"}
    tot = sum(out[base + i] for i in nm)
    for i, name in nm.items():
        print(f"   {name:22s} {out[base + i] / n:10.0f} ticks  {100.0 * out[base + i] / max(tot, 1):5.1f} %")
