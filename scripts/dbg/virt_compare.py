"""Diagnostic: one training step with and without virtual depthwise tensors; prints the parameters whose gradients differ."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import test_gpu_dist as T

dev = torch.device("cuda:0")
res = []
for virt in ("0", "1"):
    os.environ["NOISE_VIRTUAL_DW"] = virt
    m = T._build(dev)
    noisy, clean, keep = T._rank_batch(0)
    m.bind_grads()
    m._grad_target = m._flat_grads
    m.attn_keep_mask = keep.to(torch.uint8).to(dev).contiguous()
    out4 = m._native_step(noisy.to(dev), clean.to(dev), m._take_keep_mask(), want_grad=True)
    torch.cuda.synchronize()
    res.append((out4.cpu().numpy().copy(), m._flat_grads.clone(), m))
print("loss", res[0][0], res[1][0])
m = res[0][2]
for (k, p), (o, n, shape) in zip(m.named_parameters(), m._param_slices):
    a, b = res[0][1][o:o + n], res[1][1][o:o + n]
    e = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-6)
    if e > 1e-4:
        print(f"{k:50s} rel {e:.3e}  |ref| {a.abs().max().item():.3e}")
