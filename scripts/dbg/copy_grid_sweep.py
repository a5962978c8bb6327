"""Persistent grid-stride copy at different grid sizes (NOISE_COPY_GRID), against the one-shot form: how the achieved
HBM rate depends on the number of iterations a workgroup makes."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from noise_amd import _lib
from noise_amd._lib import ptr

_lib.register({"noise_bench_copy": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]})
dev = torch.device("cuda:0")
n = 1 << 30
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


for mode in (0, 4, 3):
    os.environ["NOISE_COPY_MODE"] = str(mode)
    for grid in (512, 1024, 2048, 4096, 8192, 16384, 65536, 262144):
        os.environ["NOISE_COPY_GRID"] = str(grid)
        dt = timed(lambda: _lib.call("noise_bench_copy", ptr(src), ptr(dst), n, _lib.stream()))
        per = n / 4 / (256 * (4 if mode & 4 else 1)) / grid
        print(f"mode {mode} grid {grid:7d} ({per:8.1f} iterations per workgroup) {8 * n / dt / 1e9:8.1f} GB/s")
os.environ.pop("NOISE_COPY_GRID")
os.environ["NOISE_COPY_MODE"] = "8"
dt = timed(lambda: _lib.call("noise_bench_copy", ptr(src), ptr(dst), n, _lib.stream()))
print(f"one-shot (1 iteration, {n // 1024} workgroups) {8 * n / dt / 1e9:8.1f} GB/s")
for batch in (1, 2, 4, 8, 16, 32, 63):
    os.environ["NOISE_COPY_MODE"] = str(32 + batch)
    for grid in (1024, 2048):
        os.environ["NOISE_COPY_GRID"] = str(grid)
        dt = timed(lambda: _lib.call("noise_bench_copy", ptr(src), ptr(dst), n, _lib.stream()))
        assert torch.equal(src[:4096], dst[:4096]) and torch.equal(src[-4096:], dst[-4096:])
        print(f"ticket counter, {batch} x 4 KiB per ticket, grid {grid}: {8 * n / dt / 1e9:8.1f} GB/s")
os.environ["NOISE_COPY_CHUNKED"] = "1"
for mode in (0, 4):
    os.environ["NOISE_COPY_MODE"] = str(mode)
    for grid in (1024, 2048, 8192, 65536):
        os.environ["NOISE_COPY_GRID"] = str(grid)
        dt = timed(lambda: _lib.call("noise_bench_copy", ptr(src), ptr(dst), n, _lib.stream()))
        assert torch.equal(src[:4096], dst[:4096]) and torch.equal(src[-4096:], dst[-4096:])
        print(f"contiguous share per workgroup, mode {mode}, grid {grid}: {8 * n / dt / 1e9:8.1f} GB/s")
os.environ.pop("NOISE_COPY_CHUNKED")
