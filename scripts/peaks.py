"""Measured peaks of this GPU for the two rooflines: HBM streaming copy and MFMA issue rate (fp32-input and
half-precision).  Writes gpurun_out/measured_peaks.json (copied to profiles/<round>_measured_peaks.json, which bench.py cites
next to the 8 TB/s spec figure)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from noise_amd import _lib
from noise_amd._lib import ptr

_lib.register({"noise_bench_copy": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p],
               "noise_bench_mfma": [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]})
dev = torch.device("cuda:0")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


out = {}
n = 1 << 30                                     # 4 GiB source + 4 GiB destination: far beyond the 256 MiB Infinity Cache
src, dst = torch.randn(n, device=dev), torch.empty(n, device=dev)
dt = timed(lambda: _lib.call("noise_bench_copy", ptr(src), ptr(dst), n, _lib.stream()))
out["stream_copy_GBs"] = round(8 * n / dt / 1e9, 1)
# the same copy as one workgroup per 4 KiB with non-temporal loads and stores instead of a persistent grid-stride loop
os.environ["NOISE_COPY_MODE"] = "11"
dt = timed(lambda: _lib.call("noise_bench_copy", ptr(src), ptr(dst), n, _lib.stream()))
os.environ.pop("NOISE_COPY_MODE")
out["stream_copy_oneshot_GBs"] = round(8 * n / dt / 1e9, 1)
dt = timed(lambda: torch.add(src, dst, out=dst))
out["torch_add_2r1w_GBs"] = round(12 * n / dt / 1e9, 1)
dt = timed(lambda: dst.fill_(1.0))
out["fill_GBs"] = round(4 * n / dt / 1e9, 1)
del src, dst
blocks, iters = 256 * 8, 4096                   # 8 workgroups of 4 waves per CU: 8 waves per SIMD
buf = torch.empty(blocks * 256, device=dev)
for kind, name, flop in ((0, "mfma_f32_16x16x4_TFLOPs", 2048), (1, "mfma_f16_16x16x32_TFLOPs", 16384),
                         (2, "mfma_f32_32x32x2_TFLOPs", 4096)):
    dt = timed(lambda: _lib.call("noise_bench_mfma", kind, ptr(buf), blocks, iters, _lib.stream()), reps=5)
    out[name] = round(blocks * 4 * iters * 4 * flop / dt / 1e12, 1)
out["note"] = ("copy: float4 persistent grid-stride kernel (2048 workgroups) over 2 x 4 GiB; copy_oneshot: one workgroup per 4 KiB, non-temporal loads and stores; torch_add: torch.add(a, b, out=b) on the same buffers; fill: torch fill_ of 4 GiB; mfma: 8 waves per SIMD, 4 independent "
               "accumulators per wave, random-free constant operands (clock holds higher than on random data)")
print(json.dumps(out))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "measured_peaks.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
