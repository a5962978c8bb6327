"""Token attention forward / backward of the generic path, with and without a keep mask (diagnostic timing)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from noise_amd import _lib, generic_train  # noqa: F401  (registers the entry points)
from noise_amd._lib import ptr
dev = torch.device("cuda:0")
B, n, H, dk = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 125, 6, 21
HD = H * dk; HDp = (HD + 3) // 4 * 4
HDp = 128
M = B * n
qkv = torch.randn(M, 3 * HDp, device=dev) * 0.5
o = torch.zeros(M, HDp, device=dev); go = torch.randn(M, HDp, device=dev)
dqkv = torch.zeros_like(qkv); stats = torch.empty(B * H * n * 3, device=dev)
keep = (torch.rand(B, H, n, n, device=dev) >= 0.1).to(torch.uint8)
base, dbase = qkv.data_ptr(), dqkv.data_ptr()
P = ctypes.c_void_p
def fwd(kp):
    _lib.call("noise_tok_attention", P(base), P(base + 4 * HDp), P(base + 8 * HDp), 3 * HDp, ptr(kp), 1.0 / 0.9, ptr(o), HDp, B, n, H, dk,
              float(dk) ** 0.5, _lib.stream(), nbytes=0)
def bwd(kp):
    _lib.call("noise_gen_attn_bwd", P(base), P(base + 4 * HDp), P(base + 8 * HDp), 3 * HDp, ptr(kp), 1.0 / 0.9, ptr(o), ptr(go), HDp,
              P(dbase), P(dbase + 4 * HDp), P(dbase + 8 * HDp), ptr(stats), B, n, H, dk, float(dk) ** 0.5, _lib.stream(), nbytes=0)
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for kp in (None, keep):
        for _ in range(2):
            fn(kp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(kp)
        e1.record(); torch.cuda.synchronize()
        print(f"{name} keep={'yes' if kp is not None else 'no '} B={B} n={n}: {e0.elapsed_time(e1) / 10:.3f} ms")
