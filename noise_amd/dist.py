"""Single-node data parallelism: one process per GPU, one RCCL all-reduce of the flat gradient
buffer per step (replaces the DistributedDataParallel that Lightning's `strategy: auto` builds for
the reference, train/configs/spectralv2.yaml:5-8,65,70).  BatchNorm statistics stay per rank
(sync_batchnorm: false) and buffers are never broadcast: they are constants or per-rank state.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # NOISE_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a single-GPU box)
            backend = os.environ.get("NOISE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def make_grad_all_reduce(group=None, single_rank=False):
    """Returns f(flat_grads) -> scale: sums the flat gradient over ranks in place and returns the 1/world
    factor that the fused optimizer folds into its update (so the mean is never materialised).  A group of one rank
    needs no exchange and gets None, unless `single_rank` asks for the collective anyway (a one-GPU check that the
    RCCL call path works on this machine)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not single_rank):
        return None
    world = dist.get_world_size(group)
    via_host = dist.get_backend(group) == "gloo"      # rehearsal backend: stage device tensors through the host

    def all_reduce(flat):
        if via_host and flat.is_cuda:
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world
    return all_reduce


def make_grad_mean(group=None, single_rank=False):
    """Returns f(list of gradient tensors) that replaces every tensor by its mean over ranks through ONE all-reduce of a
    flat copy (the gradients of the shape-generic training step are separate tensors, not views of one buffer), or None
    when there is nothing to exchange.  Assign it to `model.grad_sync`."""
    all_reduce = make_grad_all_reduce(group, single_rank)
    if all_reduce is None:
        return None

    def grad_mean(grads):
        flat = torch.cat([g.reshape(-1) for g in grads])
        scale = all_reduce(flat)
        flat.mul_(scale)
        o = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[o:o + n].view(g.shape))
            o += n
    return grad_mean


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) share of n_items for this rank (no data-path collective needed)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_metrics(vec, group=None):
    """Mean of a small metric vector over ranks (the epoch-end `log_dict(sync_dist=True)` of the reference)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
        vec /= dist.get_world_size(group)
    return vec
