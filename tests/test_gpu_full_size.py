"""GPU parity at the benchmark's full sizes, through properties that do not need the CPU oracle to run that size.

The small cases are pinned directly (golden vectors, oracle); these tests tie the full-size launches to them:

* the inference forward treats clips independently (eval-mode BatchNorm, attention within a clip), so a clip's output
  inside a 256 x 10 s batch must equal its output in a batch of two, which `test_forward_full_clips_vs_oracle` /
  `test_fused_forward_vs_oracle` pin against the oracle at that length;
* a training batch that repeats a small batch k times has the small batch's BatchNorm statistics (mean and biased
  variance), the same means in all three losses (train/models.py:766-804: the waveform L1 is a mean over all pairs) and
  therefore the same loss and the same parameter gradients, so the 1280-segment step must reproduce the 32-segment
  step and the CPU oracle's gradients for those 32 segments;
* the mixer works row by row (process/audiolib.py:176-231): a row's result inside 1024 x 10 s equals its result alone.
"""
import numpy as np
import pytest
import torch

from test_gpu_model import _clip, _oracle_grads, build, dev, rel  # noqa: F401  (dev is a fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,n_fft,hop", [("A", 320, 160), ("B", 512, 128)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_forward_256_clips_is_clipwise(dev, golden_dir, tag, n_fft, hop, mode):
    m = build(n_fft, hop, dev, golden_dir, max_frames=1300).eval()
    m.inference_dtype = mode
    w = _clip(256, 160000, 11).to(dev)
    w[5] = 0                                      # a silent clip inside the batch
    w[17] *= 8                                    # and a loud, partly clipped one
    w.clamp_(-1, 1)
    with torch.no_grad():
        full = m(w)
        assert full.shape == w.shape and bool(torch.isfinite(full).all())
        for idx in ([0, 5], [17, 255], [128, 129]):
            sub = m(w[idx].contiguous())
            # same kernels, same per-clip arithmetic: bit for bit
            assert torch.equal(sub, full[idx]), (idx, float((sub - full[idx]).abs().max()))
    # round trip of the spectral front end at this size: iSTFT(STFT(x)) = x away from the clip edges
    lm, ph = m.stft.forward(w)
    back = m.stft.inverse(lm, ph)
    assert back.shape[-1] == w.shape[-1]
    live = [i for i in range(256) if i != 5]      # all bins of the silent clip sit on the 1e-3 magnitude floor
    err = (back.view(256, -1) - w.view(256, -1))[live][:, n_fft:-n_fft].abs().max().item()
    assert err < 2e-5, err


def test_train_step_1280_segments_equals_replicated_small_batch(dev, golden_dir):
    B0, K = 32, 40                                # 32 segments repeated 40 times = the benchmark's 1280 segments
    gen = torch.Generator().manual_seed(77)
    clean0 = (torch.randn(B0, 8000, generator=gen) * 0.05).clamp(-1, 1)
    clean0[3] = 0                                 # a silent segment (label 0 of the activity head)
    noisy0 = (clean0 + 0.03 * torch.randn(B0, 8000, generator=gen)).clamp(-1, 1)
    keep0 = (torch.rand(B0, 6, 55, 55, generator=gen) > 0.1).to(torch.uint8)

    def step(noisy, clean, keep):
        m = build(320, 160, dev, golden_dir).train()
        with torch.no_grad():                     # well-conditioned problem, as in the oracle comparison
            m.model.decoder[-1][0].weight.mul_(0.02)
            m.model.decoder[-1][0].bias.zero_()
        opt = m.configure_optimizers()
        m.attn_keep_mask = keep.to(dev).contiguous()
        out4 = m.train_step_fused(noisy.to(dev), clean.to(dev), opt).cpu().numpy()
        grads = {k: p.grad.detach().cpu().double().numpy() for k, p in m.named_parameters()}
        params = {k: p.detach().cpu().numpy() for k, p in m.named_parameters()}
        means = {k: b.detach().cpu().numpy() for k, b in m.named_buffers() if k.endswith("running_mean")}
        return out4, grads, params, means

    small = step(noisy0, clean0, keep0)
    # the benchmark's layout: 64 clips of 10 s, cut into 20 segments each by the step itself
    big = step(noisy0.repeat(K, 1).view(64, 160000), clean0.repeat(K, 1).view(64, 160000), keep0.repeat(K, 1, 1, 1))
    np.testing.assert_allclose(big[0], small[0], rtol=2e-5, err_msg="loss, vad, waveform, spectral")
    # ... and the float64 oracle's step on the 32 segments, to the bar of the small-batch comparison: 1e-4 on the
    # loss; per parameter 2e-3 of its largest gradient entry, or twice the fp32 oracle's own distance to float64
    # where the sum behind a gradient cancels (a few BatchNorm biases with gradients near 1e-4)
    m0 = build(320, 160, dev, golden_dir)
    with torch.no_grad():
        m0.model.decoder[-1][0].weight.mul_(0.02)
        m0.model.decoder[-1][0].bias.zero_()
    sd = {k: v.detach().cpu().clone() for k, v in m0.state_dict().items()}
    args = (noisy0.view(B0, 1, 1, 8000), clean0.view(B0, 1, 1, 8000), keep0.bool())
    _, g32 = _oracle_grads(sd, *args, torch.float32)
    loss64, g64 = _oracle_grads(sd, *args, torch.float64)
    np.testing.assert_allclose(big[0][0], loss64, rtol=1e-4)
    errs = []
    for k, ref in g64.items():
        scale = max(np.abs(ref).max(), 1e-4)
        e_gpu = np.abs(big[1][k] - ref).max() / scale
        e_cpu = np.abs(g32[k] - ref).max() / scale
        errs.append((e_gpu / max(2e-3, 2.0 * e_cpu), k, e_gpu, e_cpu))
    errs.sort(reverse=True)
    assert errs[0][0] < 1.0, errs[:4]
    for k, v in small[3].items():                 # the batch mean does not change under repetition
        np.testing.assert_allclose(big[3][k], v, rtol=1e-4, atol=1e-6, err_msg=k)
    for k, p0 in small[2].items():                # AdamW's first update is lr * sign(g) where |g| is well above eps
        solid = np.abs(small[1][k]) > 1e-5
        np.testing.assert_allclose(big[2][k][solid], p0[solid], atol=5e-6, err_msg=k)


def test_mixer_1024_rows_is_rowwise(dev):
    from noise_amd import mixer
    rng = np.random.RandomState(5)
    U, L = 1024, 160000
    clean = torch.from_numpy(rng.randn(U, L) * 0.1 * rng.rand(U, 1)).to(dev)
    noise = torch.from_numpy(rng.randn(U, L) * 0.05).to(dev)
    clean[9, ::4001] *= 400                       # forces the de-clipping branch in one row
    snr = torch.from_numpy(rng.randint(-5, 20, size=U)).to(dev)
    level = torch.from_numpy(rng.randint(-35, -15, size=U)).to(dev)
    c, n, y, lv = mixer.mix_batch(clean, noise, snr, level)
    assert bool(torch.isfinite(y).all())
    np.testing.assert_allclose((c + n).cpu().numpy(), y.cpu().numpy(), rtol=0, atol=1e-15)
    assert int(lv[9].item()) != int(level[9].item())
    for idx in ([0, 9], [1022, 1023]):
        c1, n1, y1, lv1 = mixer.mix_batch(clean[idx].contiguous(), noise[idx].contiguous(), snr[idx].contiguous(),
                                          level[idx].contiguous())
        assert torch.equal(c1, c[idx]) and torch.equal(n1, n[idx]) and torch.equal(y1, y[idx])
        assert torch.equal(lv1, lv[idx])


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_streaming_1024_streams_is_streamwise(dev, golden_dir, mode):
    """The window driver (demo/src/worker.js:5-66) at the benchmark's 1024 concurrent streams: a stream's emitted
    blocks inside the batch equal the blocks of the same stream driven in a batch of two (the small driver is the
    one pinned against the oracle), over three hops, graph replay included."""
    from noise_amd import StreamingDenoiser
    m = build(320, 160, dev, golden_dir).eval()
    with torch.no_grad():
        m.model.vad.bias.fill_(3.0)               # keep the VAD gate (models.py:754) open
    m.inference_dtype = mode
    N, hops = 1024, 3
    gen = torch.Generator().manual_seed(21)
    audio = (torch.randn(N, hops * 3200, generator=gen) * 0.1).clamp(-1, 1).to(dev)
    pick = [7, 1023]
    big, small = StreamingDenoiser(m, N), StreamingDenoiser(m, len(pick))
    heard = False
    for h in range(hops):
        block = audio[:, h * 3200:(h + 1) * 3200]
        out_big = big.push_block(block.contiguous()).clone()
        out_small = small.push_block(block[pick].contiguous()).clone()
        assert out_big.shape == (N, 3200) and bool(torch.isfinite(out_big).all())
        assert torch.equal(out_big[pick], out_small), float((out_big[pick] - out_small).abs().max())
        heard = heard or float(out_big.abs().max()) > 1e-3
    assert heard
