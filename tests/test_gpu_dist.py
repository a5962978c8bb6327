"""Data-parallel training step on the GPU: gradient mean over ranks, per-rank BatchNorm statistics, no buffer
broadcast (the DDP the reference gets from Lightning's `strategy: auto`, train/configs/spectralv2.yaml:5-8,65,70).

Ranks are rehearsed on ONE card: (a) as two and as four model replicas in one process ("virtual ranks") checked against
the CPU oracle's mean gradient over the ranks' batches + AdamW, (b) as two real processes whose flat gradient buffers go
through torch.distributed (gloo, staged through the host), checked against (a)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LAYER_CONFIG = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [3, 24, 1, [3, 3], [2, 2]],
                [3, 32, 1, [3, 3], [2, 2]], [3, 64, 1, [3, 3], [2, 2]]]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
B = 6


def _build(dev):
    from noise_amd import VADSpectralV2
    m = VADSpectralV2(n_fft=320, hop_length=160, win_length=320, window="hann", nhead=6, num_layers=1, dropout=0.0,
                      layer_config=[list(r) for r in LAYER_CONFIG], pad_value=-1, max_frames=400)
    g = np.load(os.path.join(GOLDEN, "g2_state_dict.npz"))
    m.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files}, strict=False)
    with torch.no_grad():      # a well-conditioned loss (few samples on the output clamp): tight fp32 tolerances mean something
        m.model.decoder[-1][0].weight.mul_(0.02)
        m.model.decoder[-1][0].bias.zero_()
    return m.to(dev).train()


def _rank_batch(rank):
    from noise_amd.data import synth_batch
    noisy, clean, _ = synth_batch(B, 8000, "cpu", seed=500 + rank)
    keep = torch.rand(B, 6, 55, 55, generator=torch.Generator().manual_seed(900 + rank)) > 0.1
    return noisy, clean, keep


def _virtual_rank_step(dev, world=2):
    """`world` replicas, one step each into their own flat gradient buffers, summed, AdamW with grad_scale 1/world."""
    reps = [_build(dev) for _ in range(world)]
    opts = [m.configure_optimizers() for m in reps]
    losses = []
    for r, m in enumerate(reps):
        noisy, clean, keep = _rank_batch(r)
        m.bind_grads()
        m._grad_target = m._flat_grads
        m.attn_keep_mask = keep.to(torch.uint8).to(dev).contiguous()
        out4 = m._native_step(noisy.to(dev), clean.to(dev), m._take_keep_mask(), want_grad=True)
        losses.append(out4.cpu().numpy().copy())
    total = sum(m._flat_grads for m in reps[1:]) + reps[0]._flat_grads
    for m, opt in zip(reps, opts):
        m._flat_grads.copy_(total)
        opt.step(grad_scale=1.0 / world)
    return reps, total, losses


def _virtual_two_rank_step(dev):
    return _virtual_rank_step(dev, 2)


def _oracle_rank(sd, rank, dtype):
    from oracle import spectral as osp
    noisy, clean, keep = _rank_batch(rank)
    cast = (lambda v: v.to(dtype) if v.is_floating_point() else v)
    sdc = {k: cast(v) for k, v in sd.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sdc.items() if v.is_floating_point() and
              not ("running" in k or k.startswith("stft."))}
    full = dict(sdc)
    full.update(params)
    bufs = tuple(b.to(dtype) for b in osp.stft_buffers(400, 320, 160))
    stats = {}
    loss, _ = osp.shared_step(noisy.to(dtype), clean.to(dtype), full, bufs, 320, 160, training=True, new_stats=stats,
                              attn_keep=keep)
    loss.backward()
    return loss.item(), {k: v.grad.double() for k, v in params.items()}, stats


@pytest.mark.parametrize("world", [2, 4])
def test_virtual_ranks_vs_oracle(world):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import spectral as osp
    dev = torch.device("cuda:0")
    sd0 = {k: v.detach().cpu().clone() for k, v in _build(dev).state_dict().items()}
    reps, total, losses = _virtual_rank_step(dev, world)
    o64 = [_oracle_rank(sd0, r, torch.float64) for r in range(world)]
    o32 = [_oracle_rank(sd0, r, torch.float32) for r in range(world)]
    names = [k for k, _ in reps[0].named_parameters()]
    slices = reps[0]._param_slices
    inv = 1.0 / world
    worst = ("", 0.0)
    for (k, p), (o, n, shape) in zip(reps[0].named_parameters(), slices):
        mean_gpu = (inv * total[o:o + n]).view(shape).cpu().double().numpy()
        ref = inv * sum(o64[r][1][k] for r in range(world)).numpy()
        ref32 = inv * sum(o32[r][1][k] for r in range(world)).numpy()
        scale = max(np.abs(ref).max(), 1e-4)
        e_gpu, e_cpu = np.abs(mean_gpu - ref).max() / scale, np.abs(ref32 - ref).max() / scale
        ratio = e_gpu / max(2e-3, 2 * e_cpu)      # fp32 oracle's own distance to float64 calibrates "equal"
        if ratio > worst[1]:
            worst = (k, ratio)
    assert worst[1] < 1.0, worst
    for r in range(world):
        np.testing.assert_allclose(losses[r][0], o64[r][0], rtol=1e-4)
    # the replicas took the same update from the same start: bit-identical parameters afterwards
    for m in reps[1:]:
        assert torch.equal(reps[0]._flat_params, m._flat_params)
    # ... and it is AdamW on the mean gradient
    for (k, p), (o, n, shape) in zip(reps[0].named_parameters(), slices):
        g = (inv * total[o:o + n]).view(shape).cpu()
        w = sd0[k].clone()
        osp.adamw_step(w, g, torch.zeros_like(w), torch.zeros_like(w), 1)
        np.testing.assert_allclose(p.detach().cpu().numpy(), w.numpy(), rtol=2e-6, atol=2e-7, err_msg=k)
    # BatchNorm statistics stay per rank: each replica's running stats follow ITS batch
    differs = 0
    for r, m in enumerate(reps):
        for k, b in m.named_buffers():
            if "running" in k:
                np.testing.assert_allclose(b.cpu().numpy(), o64[r][2][k].float().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
    for (k, a), (_, b) in zip(reps[0].named_buffers(), reps[1].named_buffers()):
        if "running_mean" in k:
            differs += int(not torch.equal(a, b))
    assert differs > 10
    assert names


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", NOISE_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from noise_amd import dist as ndist
    ndist.init_from_env()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    m = _build(dev)
    opt = m.configure_optimizers()
    all_reduce = ndist.make_grad_all_reduce()
    noisy, clean, keep = _rank_batch(rank)
    m.attn_keep_mask = keep.to(torch.uint8).to(dev).contiguous()
    out4 = m.train_step_fused(noisy.to(dev), clean.to(dev), opt, all_reduce)
    metrics = ndist.reduce_metrics(out4.cpu().clone())
    q.put((rank, m._flat_params.cpu().numpy(), m._flat_grads.cpu().numpy(), out4.cpu().numpy(), metrics.numpy(),
           m.model.encoder[0][1].running_mean.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_match_virtual_ranks():
    """train_step_fused with the real all-reduce closure (torch.distributed, two processes on one card)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    dev = torch.device("cuda:0")
    reps, total, losses = _virtual_two_rank_step(dev)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, params, grads, out4, metrics, bn_mean in res:
        # same kernels, same inputs, same order of summation (a + b is commutative): bit-identical to the rehearsal
        assert np.array_equal(grads, total.cpu().numpy())
        assert np.array_equal(params, reps[rank]._flat_params.cpu().numpy())
        assert np.array_equal(out4, losses[rank])
        np.testing.assert_allclose(metrics, 0.5 * (losses[0] + losses[1]), rtol=1e-6)
        assert np.array_equal(bn_mean, reps[rank].model.encoder[0][1].running_mean.cpu().numpy())
    assert not np.array_equal(res[0][5], res[1][5])      # statistics are per rank, never broadcast


def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    from noise_amd import dist as ndist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    except Exception as e:      # no RCCL on this machine: reported, not a failure of the code under test
        q.put(("unavailable", repr(e)))
        return
    noisy, clean, keep = _rank_batch(0)
    out = []
    for hook in (None, ndist.make_grad_all_reduce(single_rank=True)):
        m = _build(dev)
        opt = m.configure_optimizers()
        m.attn_keep_mask = keep.to(torch.uint8).to(dev).contiguous()
        out4 = m.train_step_fused(noisy.to(dev), clean.to(dev), opt, hook)
        torch.cuda.synchronize()
        out.append((m._flat_params.cpu().numpy(), m._flat_grads.cpu().numpy(), out4.cpu().numpy()))
    q.put(("ok", dist.get_backend(), out))
    dist.destroy_process_group()


def test_rccl_all_reduce_on_the_launch_stream():
    """The production branch of the gradient exchange (backend "nccl" = RCCL, device tensor, the kernels' own stream) in
    a group of one rank: the sum over one rank must leave the step bit-identical to a step without the hook."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=240)
    p.join(timeout=120)
    if res[0] == "unavailable":
        pytest.skip(f"RCCL process group could not be created here: {res[1]}")
    assert p.exitcode == 0 and res[1] == "nccl"
    (p0, g0, l0), (p1, g1, l1) = res[2]
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1) and np.array_equal(l0, l1)
