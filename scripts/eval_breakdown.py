import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from noise_amd import VADSpectralV2, _lib
from noise_amd.data import synth_batch
dev = torch.device("cuda:0")
n_fft, hop = (320, 160) if len(sys.argv) < 2 or sys.argv[1] == "A" else (512, 128)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
m = VADSpectralV2(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann", nhead=6, num_layers=1, dropout=0.0,
                  layer_config=[list(r) for r in bench.LAYER_CONFIG], pad_value=-1, max_frames=1300).to(dev).eval()
x, _, _ = synth_batch(B, 160000, dev, seed=7)
with torch.no_grad():
    for _ in range(2): m(x.view(B, 1, 1, -1))
    torch.cuda.synchronize()
    _lib.PROFILE = prof = []
    t0 = time.perf_counter()
    m(x.view(B, 1, 1, -1))
    torch.cuda.synchronize()
    print("wall ms", 1000 * (time.perf_counter() - t0))
    _lib.PROFILE = None
fam = {}
for name, nb, e0, e1 in prof:
    f = fam.setdefault(name, [0.0, 0, 0]); f[0] += e0.elapsed_time(e1); f[1] += nb; f[2] += 1
for name, (ms, nb, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:28s} {ms:8.3f} ms {n:3d} calls {nb/1e6/max(ms,1e-9):9.1f} GB/s")
print("sum", sum(v[0] for v in fam.values()))
