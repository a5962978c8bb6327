"""Per-call table of the generic 1x1 kernel in the waveform U-Net's eval forward (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import make_golden_wave as MG
from noise_amd import _lib, generic
from noise_amd.models import VADMobileNetV2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
net = VADMobileNetV2(**MG.kwargs(MG.YAML)).cuda().eval()
x = (torch.randn(B, 1, 1, 8000, device="cuda") * 0.1)
orig = generic.GenericForward.pw
shapes = []
def pw(self, src, w, ldw, K, N, npix):
    shapes.append((K, N, npix))
    return orig(self, src, w, ldw, K, N, npix)
generic.GenericForward.pw = pw
for _ in range(2):
    shapes.clear(); net(x)
_lib.PROFILE = prof = []
shapes.clear(); net(x); torch.cuda.synchronize(); _lib.PROFILE = None
calls = [(n, nb, a.elapsed_time(b)) for n, nb, a, b in prof if n == "noise_ofa_pw"]
for (K, N, npix), (_, nb, ms) in zip(shapes, calls):
    print(f"K {K:4d} N {N:4d} npix {npix:9d}  {ms:7.3f} ms  {nb / 1e6 / ms:8.1f} GB/s  {2 * npix * K * N / ms / 1e9:7.1f} TFLOP/s")
