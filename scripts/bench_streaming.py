"""Latency / throughput of the streaming window driver (worker.js protocol: 8000-sample window, 3200 new samples per
model run) on the fp32 and the bf16 model path, for 1 and 1024 concurrent streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from noise_amd import StreamingDenoiser, VADSpectralV2  # noqa: E402

LAYER_CONFIG = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [3, 24, 1, [3, 3], [2, 2]],
                [3, 32, 1, [3, 3], [2, 2]], [3, 64, 1, [3, 3], [2, 2]]]


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = VADSpectralV2(n_fft=320, hop_length=160, win_length=320, window="hann", nhead=6, num_layers=1, dropout=0.0,
                      layer_config=[list(r) for r in LAYER_CONFIG], pad_value=-1, max_frames=1200).to(dev).eval()
    for mode in ("fp32", "bf16"):
        m.inference_dtype = mode
        for n in (1, 1024):
            s = StreamingDenoiser(m, n)
            block = torch.randn(n, 3200, device=dev) * 0.05
            for _ in range(5):
                s.push_block(block)
            torch.cuda.synchronize()
            reps = 200 if n == 1 else 30
            lat = []
            for _ in range(reps):
                t0 = time.perf_counter()
                s.push_block(block)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            lat.sort()
            p50, p99 = lat[len(lat) // 2], lat[min(len(lat) - 1, int(len(lat) * 0.99))]
            print(f"{mode} streams={n:5d}: p50 {p50 * 1e3:.3f} ms  p99 {p99 * 1e3:.3f} ms per hop  "
                  f"{n * 0.2 / p50:,.0f} audio-s/s")
            del s
            m.model._plans.clear()


if __name__ == "__main__":
    main()
