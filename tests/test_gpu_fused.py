"""GPU parity of the block-fused bf16 inference kernels against the fp32 CPU oracle.

bf16 storage with fp32 accumulation: every activation that leaves a kernel, the hidden tiles inside a kernel and the
folded weights are rounded to 8 significant bits (relative step 2^-8), so the bar is a stated bf16 tolerance, not
the fp32 one: per block, root-mean-square error <= 1 % and largest error <= 4 % of the largest reference value."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _randomise_bn(mod, gen):
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=gen))
                m.running_mean.copy_(0.2 * torch.randn(m.bias.shape, generator=gen))
                m.running_var.copy_(0.5 + torch.rand(m.bias.shape, generator=gen))


def _check(got, ref, rms_tol=1e-2, max_tol=4e-2, what=""):
    got, ref = got.double(), ref.double()
    scale = ref.abs().max().item()
    rms = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    worst = ((got - ref).abs().max() / scale).item()
    assert rms < rms_tol and worst < max_tol, f"{what}: rms {rms:.3e}, max {worst:.3e}"
    return rms, worst


@pytest.mark.parametrize("cin,hid,cout,transposed,in_f32,out_f32", [
    (16, 48, 24, False, False, False), (24, 72, 32, False, False, False), (32, 96, 64, False, False, True),
    (64, 192, 32, True, True, False), (32, 96, 24, True, False, False), (24, 72, 16, True, False, False)])
@pytest.mark.parametrize("B,H,W,crop", [(2, 21, 37, (0, 0)), (1, 8, 5, (1, 2)), (3, 5, 70, (0, 3))])
def test_fused_ir_block(dev, cin, hid, cout, transposed, in_f32, out_f32, B, H, W, crop):
    """One inverted-residual block (train/models.py:348-407 with Conv2d or ConvTranspose2d depthwise) in eval mode:
    odd sizes, tiles that overhang the image, a skip tensor added with a crop offset (decoder blocks)."""
    from noise_amd import fused
    from noise_amd.models import InvertedResidual
    from oracle import spectral as osp
    gen = torch.Generator().manual_seed(cin * 1000 + H * 10 + W)
    blk = InvertedResidual(cin, cout, [3, 3], (2, 2), 3, transposed=transposed).eval()
    _randomise_bn(blk, gen)
    x = torch.randn(B, cin, H, W, generator=gen)
    skip = torch.randn(B, cin, H + crop[0], W + crop[1], generator=gen) if transposed else None
    # reference on exactly the values the kernel receives (bf16-rounded inputs when the input tensor is bf16)
    act_dtype = torch.float32 if in_f32 else torch.bfloat16
    xq = x.to(act_dtype).float()
    sq = skip.to(act_dtype).float() if skip is not None else None
    xin = xq + sq[:, :, crop[0]:, crop[1]:] if skip is not None else xq
    sd = {"b." + k: v for k, v in blk.state_dict().items()}
    ref = osp._ir_block(xin, sd, "b", dict(t=3, stride=(2, 2)), transposed, False, None)
    wts = fused.IrBlockWeights(blk.to(dev).conv, cin, hid, cout)
    xd = xq.permute(0, 2, 3, 1).contiguous().to(dev).to(act_dtype)
    sdv = sq.permute(0, 2, 3, 1).contiguous().to(dev).to(act_dtype) if skip is not None else None
    Ho, Wo = ref.shape[2], ref.shape[3]
    out = torch.full((B, Ho, Wo, cout), float("nan"), device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
    fused.ir_block(xd, sdv, crop, wts, out, transposed)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    _check(got, ref, what=f"ir block {cin}->{hid}->{cout}")
