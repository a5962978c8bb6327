"""Throughput of the on-GPU waveform augmentation (train/data.py:439-461) at B pairs of 10 s clips.

usage: python scripts/bench_augment.py [B] [ir_seconds] [--cpu]
Prints per-entry-point times (HIP events on the launch stream), the whole-batch rate in audio-seconds of source clip
per second, and with --cpu the numpy oracle's rate on a bounded sample of the same items (one core)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from noise_amd import _lib, augment  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(args[0]) if args else 256
    ir_s = float(args[1]) if len(args) > 1 else 1.0
    L, length = 160000, 8000
    rng = np.random.RandomState(0)
    N = 64
    clean = (rng.randn(N, L) * 0.05).astype(np.float32)
    noises = [(rng.randn(L * 3) * 0.05).astype(np.float32), (rng.randn(48000) * 0.1).astype(np.float32)]
    n_ir = int(ir_s * 16000)
    irs = [(rng.randn(n_ir) * np.exp(-np.arange(n_ir) / (0.2 * n_ir))).astype(np.float32) for _ in range(8)]
    pools = augment.AugmentPools(clean, noises, irs)
    t = augment.VADTrainAugment(pools, length, p=0.5, seed=0)
    idx = [i % N for i in range(B)]
    items = t.draw(idx)
    for _ in range(3):
        t.apply(items)
    torch.cuda.synchronize()
    _lib.PROFILE = []
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        t.apply(items)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    prof, _lib.PROFILE = _lib.PROFILE, None
    agg = {}
    for name, nbytes, e0, e1 in prof:
        a = agg.setdefault(name, [0.0, 0, 0])
        a[0] += e0.elapsed_time(e1)
        a[1] += nbytes
        a[2] += 1
    n_rev = int((items["ir"] >= 0).sum())
    print(f"B={B} pairs x {L / 16000:.0f} s, {n_rev} with a {ir_s:.2f} s impulse response, output {length} samples")
    gpu_ms = 0.0
    for name, (ms, nbytes, calls) in agg.items():
        gpu_ms += ms / reps
        print(f"  {name:20s} {ms / reps:8.3f} ms/batch  {nbytes / reps / 1e6:9.1f} MB  {nbytes / ms / 1e6:8.1f} GB/s")
    out = {"pairs": B, "with_ir": n_rev, "ir_seconds": ir_s, "gpu_ms_per_batch": gpu_ms, "wall_ms_per_batch": wall * 1e3,
           "audio_s_per_s_gpu": B * L / 16000 / (gpu_ms * 1e-3), "audio_s_per_s_wall": B * L / 16000 / wall}
    if "--cpu" in sys.argv:
        from oracle import augment as oa
        noise_pool = np.concatenate(noises)
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 10.0 and n < B:
            oa.vad_pair(clean[items[n]["src"]], items[n], noise_pool, irs, length)
            n += 1
        dt = time.perf_counter() - t0
        out["cpu_oracle_audio_s_per_s_1core"] = n * L / 16000 / dt
        out["cpu_sample_pairs"] = n
    print(json.dumps(out))


if __name__ == "__main__":
    main()
