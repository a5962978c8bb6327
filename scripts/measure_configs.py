"""Throughput of the other BASELINE configurations on one MI355X (for the design notes; bench.py measures the
training step): eval forward of 10 s clips at A (320/160) and B (512/128), STFT feature generation, mixing."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from noise_amd import VADSpectralV2, mixer
from noise_amd.data import synth_batch

dev = torch.device("cuda:0")


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
for tag, n_fft, hop in (("A", 320, 160), ("B", 512, 128)):
    torch.manual_seed(0)
    m = VADSpectralV2(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann", nhead=6, num_layers=1, dropout=0.0,
                      layer_config=[list(r) for r in bench.LAYER_CONFIG], pad_value=-1, max_frames=1300).to(dev).eval()
    for B in (8, 64, 256):
        x, _, _ = synth_batch(B, 160000, dev, seed=7)
        with torch.no_grad():
            dt = timed(lambda: m(x.view(B, 1, 1, -1)), reps=3)
        out[f"forward_{tag}_B{B}x10s"] = round(B * 10 / dt, 1)
    x, _, _ = synth_batch(256, 160000, dev, seed=7)
    with torch.no_grad():
        dt = timed(lambda: m.stft.forward(x.view(256, 1, 1, -1)))
    out[f"stft_{tag}_B256x10s"] = round(2560 / dt, 1)
    del m
    torch.cuda.empty_cache()
# streaming window: (N,1,1,8000) forward, 0.2 s of new audio per stream per step
torch.manual_seed(0)
m = bench.build_model(dev).eval()
for N in (1, 1024):
    x, _, _ = synth_batch(N, 8000, dev, seed=9)
    with torch.no_grad():
        dt = timed(lambda: m(x.view(N, 1, 1, -1)), reps=20)
    out[f"window_forward_A_N{N}"] = {"ms_per_step": round(1000 * dt, 3), "audio_s_per_s": round(N * 0.2 / dt, 1)}
    run = m.graphed_forward(x.view(N, 1, 1, -1))
    ref = m(x.view(N, 1, 1, -1)).clone()
    got = run(x.view(N, 1, 1, -1))
    torch.cuda.synchronize()
    assert torch.equal(ref, got), "graph replay differs from the eager forward"
    dt = timed(lambda: run(x.view(N, 1, 1, -1)), reps=50)
    out[f"window_forward_A_N{N}_hipgraph"] = {"ms_per_step": round(1000 * dt, 3), "audio_s_per_s": round(N * 0.2 / dt, 1)}
# streaming driver (worker.js protocol, 0.2 s of new audio per stream and hop): latency per hop at N = 1 over 1000 hops
# (host-synchronised, HIP-graph replay) and throughput at N = 1024
from noise_amd import StreamingDenoiser
import time as _time
for N in (1, 1024):
    sdrv = StreamingDenoiser(m, N)
    blk = synth_batch(N, 3200, dev, seed=10)[0].view(N, 3200)
    for _ in range(5):
        sdrv.push_block(blk)
    torch.cuda.synchronize()
    lat = []
    for _ in range(1000 if N == 1 else 50):
        t0 = _time.perf_counter()
        sdrv.push_block(blk)
        torch.cuda.synchronize()
        lat.append(_time.perf_counter() - t0)
    lat.sort()
    p50, p99 = lat[len(lat) // 2], lat[int(len(lat) * 0.99) - 1]
    out[f"stream_driver_A_N{N}"] = {"hop_ms_p50": round(1000 * p50, 3), "hop_ms_p99": round(1000 * p99, 3),
                                    "audio_s_per_s": round(N * 0.2 / p50, 1)}
# mixing: U utterances of 10 s, float64
U, L = 1024, 160000
g = torch.Generator(device="cpu").manual_seed(3)
clean = (torch.randn(U, L, generator=g, dtype=torch.float64) * 0.05).to(dev)
noise = (torch.randn(U, L, generator=g, dtype=torch.float64) * 0.02).to(dev)
snr = torch.randint(-5, 20, (U,), generator=g).to(dev)
level = torch.randint(-35, -15, (U,), generator=g).to(dev)
dt = timed(lambda: mixer.mix_batch(clean, noise, snr, level))
out["mix_1024x10s_f64"] = {"audio_s_per_s": round(U * 10 / dt, 1), "GB_per_s": round(56 * L * U / dt / 1e9, 1)}
# sharded synthesis driver: mix + STFT features at both configurations, 1024 utterances of 10 s on one rank
from noise_amd import STFT, synth
sts = [STFT(1200, 320, 160, 320, "hann").to(dev), STFT(1300, 512, 128, 512, "hann").to(dev)]
synth.generate_shard(256, 0, 1, block=256, stfts=sts)          # warm-up
torch.cuda.synchronize()
t0 = _time.perf_counter()
st_ = synth.generate_shard(1024, 0, 1, block=256, stfts=sts)
torch.cuda.synchronize()
dt = _time.perf_counter() - t0
out["synth_mix_plus_stftAB_1024x10s"] = {"audio_s_per_s": round(st_["audio_seconds"] / dt, 1), "kept": st_["kept"],
                                         "note": "includes generating the synthetic source audio on the GPU"}
print(json.dumps(out, indent=1))
