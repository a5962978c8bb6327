"""configs[4]: sharded synthesis (SNR mix + clip check + STFT features at 320/160 and 512/128) on audio that is already
resident on the GPU, so the timed region holds the kernels of the path and nothing else (no synthetic-audio RNG).

    python scripts/bench_synth.py [n_files]        # prints the per-entry-point table and one JSON line
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_amd import STFT, _lib, synth

dev = torch.device("cuda:0")
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
block = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = 160000
# "real audio": four blocks of utterance pairs prepared once (a caller with wav files fills them from disk)
pool = [synth.clean_noise_block(b, block, L, 0, dev)[:2] for b in range(2)]


def source(b, n, length, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) << 20) + int(b))
    c, nz = pool[b % len(pool)]
    return c[:n], nz[:n], g


sts = [STFT(1200, 320, 160, 320, "hann").to(dev), STFT(1300, 512, 128, 512, "hann").to(dev)]
synth.generate_shard(n_files, 0, 1, length=L, block=block, stfts=sts, source=source)   # warm-up: allocator, code objects
torch.cuda.synchronize()
t0 = time.perf_counter()
st = synth.generate_shard(n_files, 0, 1, length=L, block=block, stfts=sts, source=source)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_lib.PROFILE = prof = []
synth.generate_shard(n_files, 0, 1, length=L, block=block, stfts=sts, source=source)
torch.cuda.synchronize()
_lib.PROFILE = None
agg = {}
for name, nbytes, e0, e1 in prof:
    a = agg.setdefault(name, [0.0, 0, 0])
    a[0] += e0.elapsed_time(e1)
    a[1] += nbytes
    a[2] += 1
for name, (ms, nb, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:24s} {ms:8.3f} ms  {n:3d} calls  {nb / 1e6:9.1f} MB  {nb / 1e6 / ms:8.1f} GB/s")
print(json.dumps({"synth_mix_plus_stftAB": {"files": n_files, "seconds": round(dt, 4),
                                            "audio_s_per_s": round(st["audio_seconds"] / dt, 1), "kept": st["kept"],
                                            "kernel_ms": round(sum(v[0] for v in agg.values()), 3)}}))
