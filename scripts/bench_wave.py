"""Waveform-domain U-Net forward (train/configs/mobilenetv2.yaml) on the shape-generic kernels: per-entry-point table, and
the CPU oracle (torch CPU kernels) on a bounded sample of the same workload.

    python scripts/bench_wave.py [clips] [samples]      # default: 1280 segments of 8000 samples (the training batch shape)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import make_golden_wave as MG  # configuration of mobilenetv2.yaml
from noise_amd import _lib
from noise_amd.models import VADMobileNetV2

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
torch.manual_seed(0)
net = VADMobileNetV2(**MG.kwargs(MG.YAML))
x = (torch.randn(B, 1, 1, L) * 0.1).clamp(-1, 1)
xd = x.to(dev)
net = net.to(dev)
for mode in ("eval", "train"):
    net.train(mode == "train")
    for _ in range(2):
        net(xd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        net(xd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    _lib.PROFILE = prof = []
    net(xd)
    torch.cuda.synchronize()
    _lib.PROFILE = None
    agg = {}
    for name, nbytes, a, b in prof:
        r = agg.setdefault(name, [0.0, 0, 0])
        r[0] += a.elapsed_time(b)
        r[1] += nbytes
        r[2] += 1
    print(f"== {mode}: {B} clips x {L} samples: {ms:.3f} ms/forward, {len(prof)} launches, {B * L / 16000 / ms * 1e3:,.0f} audio-s/s")
    for name, (t, nb, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"   {name:24s} {t:8.3f} ms  {n:4d} calls  {nb / 1e6:9.1f} MB  {nb / 1e6 / max(t, 1e-9):8.1f} GB/s")
    print(json.dumps({"wave_unet_forward": {"mode": mode, "clips": B, "samples": L, "ms": round(ms, 3),
                                            "audio_s_per_s": round(B * L / 16000 / ms * 1e3, 1)}}))

# training step (forward + losses + backward; gradients land in .grad, the optimizer is the caller's)
if L == 8000:
    net.train()
    clean = (torch.randn(B, 1, 1, L) * 0.05).clamp(-1, 1).to(dev)
    for _ in range(2):
        net.training_step((xd, clean, None), 0).backward()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        net.training_step((xd, clean, None), 0).backward()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    _lib.PROFILE = prof = []
    net.training_step((xd, clean, None), 0).backward()
    torch.cuda.synchronize()
    _lib.PROFILE = None
    agg = {}
    for name, nbytes, a, b in prof:
        r = agg.setdefault(name, [0.0, 0, 0])
        r[0] += a.elapsed_time(b)
        r[1] += nbytes
        r[2] += 1
    print(f"== training step: {B} segments x {L} samples: {ms:.3f} ms/step, {len(prof)} launches, {B * L / 16000 / ms * 1e3:,.0f} audio-s/s,"
          f" peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    for name, (t, nb, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"   {name:26s} {t:8.3f} ms  {n:4d} calls  {nb / 1e6:9.1f} MB  {nb / 1e6 / max(t, 1e-9):8.1f} GB/s")
    if os.environ.get("NOISE_WAVE_CALLS"):   # every launch of the step, in order
        for i, (name, nbytes, a, b) in enumerate(prof):
            t = a.elapsed_time(b)
            print(f"   #{i:3d} {name:26s} {t:7.3f} ms {nbytes / 1e6:9.1f} MB {nbytes / 1e6 / max(t, 1e-9):8.1f} GB/s")
    print(json.dumps({"wave_unet_train_step": {"segments": B, "samples": L, "ms": round(ms, 3),
                                               "audio_s_per_s": round(B * L / 16000 / ms * 1e3, 1)}}))

# CPU oracle on a bounded sample (test infrastructure timed as the baseline, never the product path)
from oracle import wave_unet as owu
cpu = VADMobileNetV2(**MG.kwargs(MG.YAML)).eval()
nb = max(1, min(B, 16 * 8000 // L))
xs = x[:nb]
owu.full_forward(cpu, xs, False)
t0 = time.perf_counter()
reps = 0
while time.perf_counter() - t0 < 10.0:
    owu.full_forward(cpu, xs, False)
    reps += 1
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"wave_unet_forward_cpu": {"kind": "port (CPU oracle: torch CPU kernels)", "threads": torch.get_num_threads(),
                                            "clips": nb, "samples": L, "ms": round(dt * 1e3, 1),
                                            "audio_s_per_s": round(nb * L / 16000 / dt, 1)}}))
