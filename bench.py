#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per second of the full training step
(STFT x2 -> denoiser fwd -> iSTFT -> losses -> backward -> gradient all-reduce -> AdamW) on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): per GPU a batch of 64
synthetic 16 kHz clips of 10 s, which the training step views as 1280 segments of 8000 samples
(train/models.py:768-769), reference STFT configuration n_fft 320 / hop 160, fp32.  Weak scaling: every
rank processes its own 640 audio-seconds per step; the only exchange is one all-reduce of the 76,146-float
gradient buffer.  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LAYER_CONFIG = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [3, 24, 1, [3, 3], [2, 2]],
                [3, 32, 1, [3, 3], [2, 2]], [3, 64, 1, [3, 3], [2, 2]]]
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)


def build_model(device):
    from noise_amd import VADSpectralV2
    torch.manual_seed(0)
    m = VADSpectralV2(n_fft=320, hop_length=160, win_length=320, window="hann", nhead=6, num_layers=1, dropout=0.0,
                      layer_config=[list(r) for r in LAYER_CONFIG], pad_value=-1, max_frames=1200)
    return m.to(device).train()


def cpu_baseline(seconds_budget=20.0):
    """The CPU oracle (a port of the reference algorithm) on a bounded sample of the same workload:
    training steps of 128 segments (64 audio-seconds) on the host cores."""
    from oracle import spectral as osp
    from noise_amd.data import synth_batch
    # all the host cores this job may use: the GPU box gives a one-GPU job a share of 16 cores of a much larger host and
    # exports it as OMP_NUM_THREADS (os.cpu_count() would report the whole machine)
    threads = int(os.environ.get("OMP_NUM_THREADS") or min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = build_model("cpu")
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and
              not ("running" in k or k.startswith("stft."))}
    sd.update(params)
    bufs = osp.stft_buffers(1200, 320, 160)
    B = 128
    noisy, clean, _ = synth_batch(B, 8000, "cpu", seed=1234)
    mom = {k: torch.zeros_like(p) for k, p in params.items()}
    var = {k: torch.zeros_like(p) for k, p in params.items()}
    steps, t0 = 0, time.perf_counter()
    while True:
        keep = torch.rand(B, 6, 55, 55) > 0.1
        loss, _ = osp.shared_step(noisy, clean, sd, bufs, 320, 160, training=True, new_stats={}, attn_keep=keep)
        loss.backward()
        with torch.no_grad():
            for k, p in params.items():
                osp.adamw_step(p, p.grad, mom[k], var[k], steps + 1)
                p.grad = None
        steps += 1
        el = time.perf_counter() - t0
        if steps >= 2 and (el > seconds_budget or steps >= 60):
            break
    # first step is warm-up-ish on CPU as well; report the steady average over all timed steps
    return {"value": round(steps * B * 0.5 / el, 2), "unit": "audio-sec/sec", "cores": threads, "kind": "port",
            "sample": f"{steps} training steps of {B} x 0.5 s segments (fwd+bwd+AdamW, fp32) on the CPU oracle, torch CPU "
                      f"kernels on {threads} threads = this job's share of the host's cores (OMP_NUM_THREADS)"}


# entry point -> the kernel that moves its bytes (names as rocprofv3 reports them, template arguments dropped)
MAIN_KERNEL = {"noise_pw_bwd": "pw_bwd_kernel", "noise_pw_fwd": "pw_gemm_kernel", "noise_dw3x3_bwd": "walk_gather_bwd_kernel",
               "noise_dw3x3_fwd": "walk_gather_fwd_kernel", "noise_last_convT_bwd": "last_gather_bwd_kernel",
               "noise_last_convT_fwd": "last_cell_fwd_kernel", "noise_stft_fwd": "stft_frames_kernel"}


PROFILE_ROUND = "r3"      # which committed profile set the quoted (not live-measured) figures come from


def _profile(name):
    for rnd in (PROFILE_ROUND, "r2"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
        if os.path.exists(path):
            return path
    return None


def measured_peaks():
    """What this GPU model achieved in the micro-benchmarks of scripts/peaks.py: streaming copy / fill rates and MFMA
    issue rates, quoted next to the spec peak the roofline fraction is priced against.  NOT measured in this run: read
    from the committed file named in "source"."""
    path = _profile("measured_peaks.json")
    if path is None:
        return None
    with open(path) as f:
        d = json.load(f)
    d.pop("note", None)
    d["source"] = os.path.relpath(path, ROOT) + " (committed micro-benchmark results, not measured in this run)"
    return d


def other_configs(device):
    """Short measurements of the other BASELINE configurations on this GPU, reported beside the headline number:
    configs[1] (256 x 10 s clips, bf16 block-fused inference forward STFT -> denoiser -> iSTFT) at the reference STFT
    (320/160) and at BASELINE's (512/128), and the same forward in fp32."""
    from noise_amd import VADSpectralV2
    from noise_amd.data import synth_batch
    out = {}
    B = 256
    x, _, _ = synth_batch(B, 160000, device, seed=7)
    x = x.view(B, 1, 1, -1)
    for tag, n_fft, hop in (("320_160", 320, 160), ("512_128", 512, 128)):
        torch.manual_seed(0)
        m = VADSpectralV2(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann", nhead=6, num_layers=1, dropout=0.0,
                          layer_config=[list(r) for r in LAYER_CONFIG], pad_value=-1, max_frames=1300).to(device).eval()
        for mode in ("bf16", "fp32"):
            m.inference_dtype = mode
            with torch.no_grad():
                for _ in range(2):
                    m(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    m(x)
                torch.cuda.synchronize()
            out[f"infer_{mode}_256x10s_{tag}_audio_s_per_s"] = round(5 * B * 10 / (time.perf_counter() - t0), 1)
            m.model._plans.clear()
        del m
        torch.cuda.empty_cache()
    out.update(stream_mix_augment_configs(device))
    return out


def other_configs_cpu(seconds_each=6.0):
    """CPU figures beside `other_configs`, same run, same host cores (stated): the oracle's inference forward on
    8 x 10 s clips at 320/160 and 512/128 (BASELINE configs[0]), one 0.5 s streaming window (configs[3]), the numpy
    float64 mixer of the oracle on one core (configs[4]).  Bounded samples: a few seconds each."""
    import numpy as np
    from oracle import mixer as omix
    from oracle import spectral as osp
    from noise_amd.data import synth_batch
    threads = int(os.environ.get("OMP_NUM_THREADS") or min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    out = {"cores": threads, "kind": "port (CPU oracle: torch CPU kernels / numpy)"}
    m = build_model("cpu").eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    x8 = synth_batch(8, 160000, "cpu", seed=7)[0]
    x1 = synth_batch(1, 8000, "cpu", seed=9)[0]
    for tag, n_fft, hop in (("320_160", 320, 160), ("512_128", 512, 128)):
        bufs = osp.stft_buffers(1300, n_fft, hop)
        with torch.no_grad():
            osp.model_forward(x1, sd, bufs, n_fft, hop)
            n, t0 = 0, time.perf_counter()
            while n < 1 or (time.perf_counter() - t0 < seconds_each and n < 20):
                osp.model_forward(x8, sd, bufs, n_fft, hop)
                n += 1
            dt = (time.perf_counter() - t0) / n
        out[f"infer_fp32_8x10s_{tag}_audio_s_per_s"] = round(80.0 / dt, 1)
        if tag == "320_160":
            with torch.no_grad():
                n, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < 2.0:
                    osp.model_forward(x1, sd, bufs, n_fft, hop)
                    n += 1
                dt = (time.perf_counter() - t0) / n
            out["stream_window_fp32_n1"] = {"hop_ms": round(1e3 * dt, 3), "audio_s_per_s": round(0.2 / dt, 1)}
    rng = np.random.RandomState(3)
    clean, noise = rng.randn(160000) * 0.05, rng.randn(160000) * 0.02
    n, t0 = 0, time.perf_counter()
    while n < 2 or time.perf_counter() - t0 < 3.0:
        omix.segmental_snr_mixer(clean, noise, 5, -25)
        n += 1
    dt = (time.perf_counter() - t0) / n
    out["mix_f64_10s_numpy_1core_audio_s_per_s"] = round(10.0 / dt, 1)
    return out


def stream_mix_augment_configs(device):
    """configs[3] (streaming window driver, worker.js protocol: 0.2 s of new audio per stream and hop; latency of one
    stream, throughput of 1024, bf16 model path, HIP-graph replay), configs[4] (SNR mixing of 1024 utterance pairs of
    10 s in float64) and the waveform augmentation of the data module (256 pairs of 10 s, 1 s impulse responses)."""
    import numpy as np
    from noise_amd import StreamingDenoiser, augment, mixer
    out = {}
    m = build_model(device).eval()
    m.inference_dtype = "bf16"
    for n in (1, 1024):
        s = StreamingDenoiser(m, n)
        block = torch.randn(n, 3200, device=device) * 0.05
        for _ in range(5):
            s.push_block(block)
        torch.cuda.synchronize()
        lat = []
        for _ in range(200 if n == 1 else 30):
            t0 = time.perf_counter()
            s.push_block(block)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        lat.sort()
        p50, p99 = lat[len(lat) // 2], lat[min(len(lat) - 1, int(len(lat) * 0.99))]
        out[f"stream_bf16_320_160_n{n}"] = {"hop_ms_p50": round(1e3 * p50, 3), "hop_ms_p99": round(1e3 * p99, 3),
                                            "audio_s_per_s": round(n * 0.2 / p50, 1)}
        del s
        m.model._plans.clear()
    del m
    U, L = 1024, 160000
    g = torch.Generator(device="cpu").manual_seed(3)
    clean = (torch.randn(U, L, generator=g, dtype=torch.float64) * 0.05).to(device)
    noise = (torch.randn(U, L, generator=g, dtype=torch.float64) * 0.02).to(device)
    snr = torch.randint(-5, 20, (U,), generator=g).to(device)
    level = torch.randint(-35, -15, (U,), generator=g).to(device)
    for _ in range(2):
        mixer.mix_batch(clean, noise, snr, level)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        mixer.mix_batch(clean, noise, snr, level)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out["mix_f64_1024x10s"] = {"audio_s_per_s": round(U * 10 / dt, 1), "GB_per_s": round(56 * L * U / dt / 1e9, 1)}
    del clean, noise
    torch.cuda.empty_cache()
    rng = np.random.RandomState(0)
    N, B = 64, 256
    cl = (rng.randn(N, L) * 0.05).astype(np.float32)
    irs = [(rng.randn(16000) * np.exp(-np.arange(16000) / 3200.0)).astype(np.float32) for _ in range(8)]
    pools = augment.AugmentPools(cl, [(rng.randn(3 * L) * 0.05).astype(np.float32)], irs, device)
    t = augment.VADTrainAugment(pools, 8000, p=0.5, seed=0)
    items = t.draw([i % N for i in range(B)])
    for _ in range(3):
        t.apply(items)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        t.apply(items)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out["augment_256x10s_ir1s"] = {"ms_per_batch": round(1e3 * dt, 3), "audio_s_per_s": round(B * 10 / dt, 1),
                                   "with_impulse_response": int((items["ir"] >= 0).sum())}
    return out


def pmc_traffic(entry_point, clips=64):
    """HBM bytes per launch of the entry point's main kernel, from the committed PMC passes of this same command
    (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as gfx950 requires); None when
    no such profile is available.  Returns (bytes, source file): the figure is quoted, not measured in this run."""
    path = _profile("hbm_traffic_per_kernel.json")
    kernel = MAIN_KERNEL.get(entry_point)
    if kernel is None or clips != 64 or path is None:   # the passes were taken on the default workload
        return None, None
    with open(path) as f:
        rec = json.load(f).get(kernel)
    return (int(rec["hbm_bytes_per_dispatch"]) if rec else None), os.path.relpath(path, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=64, help="10 s clips per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short inference measurements")
    ap.add_argument("--breakdown", action="store_true", help="print the per-entry-point table to stderr")
    ap.add_argument("--all-events", action="store_true", help="(diagnostic) instrument every call in the timed region too")
    args = ap.parse_args()

    from noise_amd import _lib, dist as ndist
    from noise_amd.data import synth_batch
    rank, world, local_rank = ndist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)

    model = build_model(device)
    opt = model.configure_optimizers()
    all_reduce = ndist.make_grad_all_reduce()
    clip_len = 160000
    noisy, clean, _ = synth_batch(args.clips, clip_len, device, seed=1234 + rank)
    audio_s_per_step = args.clips * clip_len / 16000.0

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def families(prof):
        fam = {}
        for name, nbytes, e0, e1 in prof:
            f = fam.setdefault(name, [0.0, 0, 0])
            f[0] += e0.elapsed_time(e1)
            f[1] += nbytes
            f[2] += 1
        return fam

    # Warm-up steps are fully instrumented (HIP events around every entry-point call on the launch stream): they
    # give the per-entry-point table and name the dominant entry point.  Bracketing all ~110 calls of a step costs
    # ~0.8 ms of host time per step, so the timed region keeps the events only around the dominant entry point:
    # its average launch duration is measured live there, and `value` is not taxed by the instrumentation.
    _lib.PROFILE = wprof = []
    for _ in range(max(args.warmup, 1)):
        model.train_step_fused(noisy, clean, opt, all_reduce)
    barrier()
    nw = max(args.warmup, 1)
    per_step = len(wprof) // nw
    if nw > 1:                       # the first step pays one-time costs (code-object loads): leave it out of the table
        wprof = wprof[per_step:]
        nw -= 1
    wfam = families(wprof)
    wtotal = sum(v[0] for v in wfam.values())
    dom = max(wfam.items(), key=lambda kv: kv[1][0])[0]
    step_bytes = sum(v[1] for v in wfam.values()) / nw
    if args.breakdown and rank == 0:
        for name, (ms, nb, n) in sorted(wfam.items(), key=lambda kv: -kv[1][0]):
            print(f"{name:28s} {ms / nw:9.3f} ms/step {100 * ms / wtotal:5.1f}%  {n // nw:4d} calls  "
                  f"{nb / 1e6 / max(ms, 1e-9):9.1f} GB/s", file=sys.stderr)
        for i, (name, nbytes, e0, e1) in enumerate(wprof[-per_step:]):    # every launch of the last warm-up step
            ms = e0.elapsed_time(e1)
            print(f"  #{i:3d} {name:26s} {ms:7.3f} ms {nbytes / 1e6:9.1f} MB {nbytes / 1e6 / max(ms, 1e-9):8.1f} GB/s",
                  file=sys.stderr)
    _lib.PROFILE = prof = []
    _lib.PROFILE_ONLY = None if args.all_events else {dom}
    # The host enqueues a step in ~2 ms and only runs ahead of the GPU after the first timed steps: a generation-2
    # garbage collection (tens of ms on torch's heap) in that window would idle the GPU and be charged to the kernels.
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out4 = model.train_step_fused(noisy, clean, opt, all_reduce)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    _lib.PROFILE = None
    _lib.PROFILE_ONLY = None
    loss = float(out4[0].item())
    if world > 1:
        on_host = torch.distributed.get_backend() == "gloo"
        t = torch.tensor([elapsed], device="cpu" if on_host else device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    dms, dbytes, dn = families(prof)[dom]
    step_ms = 1000 * elapsed / args.steps

    achieved = dbytes / 1e6 / dms  # bytes / ms -> GB/s
    traffic, traffic_src = pmc_traffic(dom, args.clips)
    roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": (traffic_src + " (PMC passes of this command, committed; not measured in this run)")
                if traffic_src else None,
                "algorithmic_bytes_per_launch": int(dbytes / dn),
                "launches_per_step": dn // args.steps, "avg_launch_ms": round(dms / dn, 4),
                "share_of_step": round(dms / args.steps / step_ms, 3),
                "step_hbm_frac": round(step_bytes / 1e6 / step_ms / HBM_PEAK_GBS, 4),
                "measured_peaks": measured_peaks()}

    if rank == 0:
        line = {
            "metric": "audio-sec/sec", "value": round(world * audio_s_per_step * args.steps / elapsed, 1),
            "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"train_step fwd+bwd+AdamW, {args.clips} x 10 s clips/GPU = {args.clips * 20} x 0.5 s segments, "
                                   "n_fft 320 hop 160",
                       "global_batch_clips": world * args.clips, "segments_per_gpu": args.clips * clip_len // 8000,
                       "parallelism": f"dp{world}", "loss": round(loss, 5)},
            "roofline": roofline,
        }
        if not args.no_other_configs and world == 1:
            del model, opt, noisy, clean
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(device)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
            if "other_configs" in line:
                line["other_configs"]["cpu"] = other_configs_cpu()
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
