"""Streaming-copy variants of the micro-benchmark (NOISE_COPY_MODE bits: 1 non-temporal stores, 2 non-temporal loads,
4 four pieces in flight per thread, 8 one-shot grid, 16 persistent with the next loads ahead of the stores) next to torch's own copy / add kernels: which access form moves the
most bytes per second on this GPU.   usage: copy_variants.py [GiB per buffer]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from noise_amd import _lib
from noise_amd._lib import ptr

_lib.register({"noise_bench_copy": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]})
dev = torch.device("cuda:0")
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * (1 << 28))
src, dst = torch.randn(n, device=dev), torch.empty(n, device=dev)


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


os.environ.pop("NOISE_COPY_MODE", None)
print(f"baseline grid-stride float4 copy: {8 * n / timed(lambda: _lib.call('noise_bench_copy', ptr(src), ptr(dst), n, _lib.stream())) / 1e9:8.1f} GB/s")
for mode in range(24):
    os.environ["NOISE_COPY_MODE"] = str(mode)
    dt = timed(lambda: _lib.call("noise_bench_copy", ptr(src), ptr(dst), n, _lib.stream()))
    assert torch.equal(src[:4096], dst[:4096]) and torch.equal(src[-4096:], dst[-4096:])
    tags = [t for b, t in ((1, "nt-store"), (2, "nt-load"), (4, "4-in-flight"), (8, "one-shot"), (16, "pipelined")) if mode & b]
    print(f"mode {mode:2d} {'+'.join(tags) or 'plain':40s} {8 * n / dt / 1e9:8.1f} GB/s")
os.environ.pop("NOISE_COPY_MODE", None)
print(f"torch copy_ : {8 * n / timed(lambda: dst.copy_(src)) / 1e9:8.1f} GB/s")
print(f"torch add   : {12 * n / timed(lambda: torch.add(src, dst, out=dst)) / 1e9:8.1f} GB/s (2R+1W)")
print(f"torch fill_ : {4 * n / timed(lambda: dst.fill_(1.0)) / 1e9:8.1f} GB/s")
