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


LAYER_CONFIG = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [3, 24, 1, [3, 3], [2, 2]],
                [3, 32, 1, [3, 3], [2, 2]], [3, 64, 1, [3, 3], [2, 2]]]


def _build(n_fft, hop, dev, golden_dir, tame=True):
    import os
    from noise_amd import VADSpectralV2
    m = VADSpectralV2(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann", nhead=6, num_layers=1, dropout=0.0,
                      layer_config=[list(r) for r in LAYER_CONFIG], pad_value=-1, max_frames=1300)
    g = np.load(os.path.join(golden_dir, "g2_state_dict.npz"))
    m.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files}, strict=False)
    gen = torch.Generator().manual_seed(5)
    _randomise_bn(m.model, gen)        # running statistics that differ from the freshly initialised (0, 1)
    if tame:
        with torch.no_grad():          # keep most of the enhanced log-magnitudes inside the [-1, 1] clamp
            m.model.decoder[-1][0].weight.mul_(0.05)
            m.model.decoder[-1][0].bias.zero_()
    return m.to(dev).eval()


def _clip(B, L, seed):
    gen = torch.Generator().manual_seed(seed)
    env = 0.5 - 0.5 * torch.cos(2 * np.pi * 4.0 * torch.arange(L) / 16000.0)
    return (0.05 * torch.randn(B, 1, 1, L, generator=gen) * env + 0.02 * torch.randn(B, 1, 1, L, generator=gen)).clamp(-1, 1)


def _snr_db(got, ref):
    got, ref = got.double(), ref.double()
    return float(10 * torch.log10(ref.pow(2).sum() / (got - ref).pow(2).sum().clamp_min(1e-30)))


@pytest.mark.parametrize("tag,n_fft,hop", [("A", 320, 160), ("B", 512, 128)])
@pytest.mark.parametrize("L,B,first_L,tame", [(8000, 3, None, True), (16000, 2, None, True), (160000, 2, None, True),
                                              (160000, 1, 8000, True), (160000, 2, None, False)])
def test_fused_forward_vs_oracle(dev, golden_dir, tag, n_fft, hop, L, B, first_L, tame):
    """Whole bf16 inference forward (STFT fp32 -> fused network -> iSTFT fp32) against the fp32 CPU oracle.
    Stated bf16 tolerance: network output (the additive mask in normalised log-magnitude) within 1 % rms / 4 % max
    of its largest value (measured: 0.7 % rms, about 25 roundings to 8 significant bits along the path); VAD logit
    within 0.03.  What that means for the waveform depends on the size of the mask, which is multiplied by
    max|log-magnitude| (about 6.9) and exponentiated: with the fixture's untrained weights as they are (|mask| up to
    ~1) the enhanced waveform keeps an SNR >= 33 dB against the fp32 oracle (measured 36 dB); with the last layer
    scaled to 5 % (`tame`, masks of the size of a small correction) >= 55 dB (measured 64-67 dB).
    `first_L`: the padding is searched on a first call of another length, so the skips are cropped (models.py:608)."""
    from oracle import spectral as osp
    m = _build(n_fft, hop, dev, golden_dir, tame)
    m.inference_dtype = "bf16"
    if first_L is not None:
        m(_clip(1, first_L, 1).to(dev))
        if tag == "B":      # 63 frames -> W_padding 2; 1009 + 2 = 1011 columns -> 506 at half resolution, 505 coming back
            L = 128 * 1008  # up: the two full-resolution skips are cropped by one column
    padding = (m.model.H_padding, m.model.W_padding) if first_L is not None else None
    w = _clip(B, L, L + B)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    bufs = osp.stft_buffers(1300, n_fft, hop)
    lm, ph = osp.stft_forward(w, bufs[0], n_fft, hop)
    mx = lm.abs().reshape(B, 1, -1).max(dim=-1, keepdim=True)[0].unsqueeze(2) + 1e-6
    filt_ref, vad_ref = osp.denoiser(lm / mx, sd, padding=padding)
    out_ref = osp.model_forward(w, sd, bufs, n_fft, hop, padding=padding)
    wd = w.to(dev)
    out = m(wd)
    lm_d, _, amax = m.stft.forward(wd, return_absmax=True)
    H, W = lm_d.shape[2:]
    plan = m.model.plan(B, H, W, dev, fused=True)
    enh, vad, filt = plan.forward(lm_d.view(B, H, W), amax, want_filt=True)
    rms, worst = _check(filt.cpu().view(B, 1, H, W), filt_ref, what=f"mask {tag} L={L}")
    assert (vad.cpu().view(-1) - vad_ref.view(-1)).abs().max() < 0.03
    snr = _snr_db(out.cpu(), out_ref)
    print(f"{tag} L={L} B={B} crop={first_L} tame={tame}: mask rms {rms:.2e} max {worst:.2e}  wave SNR {snr:.1f} dB")
    assert snr >= (55.0 if tame else 33.0)
    # fp32 mode of the same model stays on the per-layer kernels and the tight tolerance
    m.inference_dtype = "fp32"
    np.testing.assert_allclose(m(wd).cpu().numpy(), out_ref.numpy(), atol=1e-4)


def test_fused_weights_follow_training(dev, golden_dir):
    """bf16 eval, one fused training step (parameters and BatchNorm statistics change through raw device pointers,
    which bumps no tensor version), bf16 eval again: the folded weight images must be rebuilt.  The fp32 path reads
    the live parameters, so the two paths have to agree after the step as they did before it; a captured bf16 graph
    holds the old folded copies and has to refuse."""
    from noise_amd.data import synth_batch
    m = _build(320, 160, dev, golden_dir, True)
    w = _clip(2, 8000, 5).to(dev)
    m.inference_dtype = "bf16"
    before = m(w).clone()
    run = m.graphed_forward(w)
    assert torch.equal(run(w), before)
    opt = m.configure_optimizers()
    opt.lr = 1e-2                      # a step large enough to move the output far beyond the bf16 tolerance
    noisy, clean, _ = synth_batch(8, 8000, dev, seed=3)
    m.train()
    m.train_step_fused(noisy, clean, opt)
    m.eval()
    after = m(w).clone()
    m.inference_dtype = "fp32"
    ref = m(w)
    print(f"bf16 after the step vs fp32: {_snr_db(after.cpu(), ref.cpu()):.1f} dB; stale output vs fp32: "
          f"{_snr_db(before.cpu(), ref.cpu()):.1f} dB")
    assert _snr_db(after.cpu(), ref.cpu()) >= 33.0, "bf16 forward ran on stale folded weights after a training step"
    assert _snr_db(before.cpu(), ref.cpu()) < 25.0     # the step did change the output, so the check above is not vacuous
    with pytest.raises(RuntimeError, match="capture again"):
        run(w)
    # a checkpoint loaded in place is seen as well
    m.inference_dtype = "bf16"
    sd = {k: (v * 0.5 if k.endswith("decoder.4.0.weight") else v.clone()) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.inference_dtype = "fp32"
    ref2 = m(w).clone()
    m.inference_dtype = "bf16"
    assert _snr_db(m(w).cpu(), ref2.cpu()) >= 33.0


@pytest.mark.parametrize("B,n,heads,dk", [(2, 55, 6, 10), (3, 704, 6, 10), (1, 1360, 6, 10), (2, 33, 4, 16), (2, 130, 8, 7),
                                          (1, 2000, 6, 10)])
def test_attention_core_f16(dev, B, n, heads, dk):
    """Half-precision attention core of the bf16 inference path (train/models.py:27-37, eval mode) against torch:
    tolerance 2e-3 of the largest output (operands carry 11 significant bits, scores and sums are fp32)."""
    from noise_amd import _lib, fused  # noqa: F401
    from noise_amd._lib import ptr
    g = torch.Generator().manual_seed(n)
    hd = heads * dk
    qkv = torch.randn(B * n, 3 * hd, generator=g)
    q, k, v = (t.reshape(B, n, heads, dk).transpose(1, 2) for t in qkv.split(hd, dim=1))
    att = torch.softmax(torch.matmul(q / dk ** 0.5, k.transpose(2, 3)), dim=-1)
    ref = torch.matmul(att, v).transpose(1, 2).reshape(B * n, hd)
    qkv_d = qkv.to(dev)
    o = torch.full((B * n, hd), float("nan"), device=dev)
    _lib.call("noise_attn_core_fwd_f16", ptr(qkv_d), ptr(o), B * n, n, 64, heads, dk, _lib.stream())
    err = ((o.cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-3, err
