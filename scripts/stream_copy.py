"""Achievable HBM bandwidth on this GPU for the access mixes of the step: device copy (1 read + 1 write), read-only
reduction, fill (write only).  Reported as bytes moved / time."""
import torch
dev = torch.device("cuda:0")
n = 256 * 1024 * 1024          # 1 GiB of fp32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
a.normal_()


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 1e3


print(f"copy  (1R+1W): {2 * 4 * n / timed(lambda: b.copy_(a)) / 1e9:8.1f} GB/s")
print(f"sum   (1R)   : {4 * n / timed(lambda: a.sum()) / 1e9:8.1f} GB/s")
print(f"fill  (1W)   : {4 * n / timed(lambda: b.fill_(1.0)) / 1e9:8.1f} GB/s")
print(f"add   (2R+1W): {3 * 4 * n / timed(lambda: torch.add(a, b, out=b)) / 1e9:8.1f} GB/s")
