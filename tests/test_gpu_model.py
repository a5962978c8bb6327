"""GPU parity of the whole model: eval forward, training step (loss, gradients, BatchNorm state,
AdamW) against the golden vectors generated from the reference and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LAYER_CONFIG = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [3, 24, 1, [3, 3], [2, 2]],
                [3, 32, 1, [3, 3], [2, 2]], [3, 64, 1, [3, 3], [2, 2]]]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def build(n_fft, hop, dev, golden_dir, max_frames=1200):
    from noise_amd import VADSpectralV2
    lc = [list(r) for r in LAYER_CONFIG]
    m = VADSpectralV2(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann", nhead=6, num_layers=1, dropout=0.0,
                      layer_config=lc, pad_value=-1, max_frames=max_frames)
    assert lc == LAYER_CONFIG, "the caller's layer_config must not be mutated"
    g = np.load(os.path.join(golden_dir, "g2_state_dict.npz"))
    sd = {k: torch.from_numpy(g[k]) for k in g.files}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("stft.") for k in missing)
    return m.to(dev)


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


@pytest.mark.parametrize("tag,n_fft,hop", [("A", 320, 160), ("B", 512, 128)])
def test_eval_forward_golden(dev, golden_dir, tag, n_fft, hop):
    g = np.load(os.path.join(golden_dir, f"g2_eval_forward_{tag}.npz"))
    for s in ("seg", "long"):
        m = build(n_fft, hop, dev, golden_dir).eval()
        w = torch.from_numpy(g[f"{s}_wave"]).to(dev)
        out = m(w)
        assert [m.model.H_padding, m.model.W_padding] == list(g[f"{s}_padding"])
        assert out.shape == g[f"{s}_out"].shape
        # fp32 end to end: 1e-4 relative on the enhanced log-magnitudes, 1e-4 absolute on the [-1,1] waveform
        lm, ph = m.stft.forward(w)
        enh, vad = m._forward(lm)
        assert rel(enh.cpu().numpy(), g[f"{s}_enh_logmag"]) < 1e-4
        np.testing.assert_allclose(vad.cpu().numpy(), g[f"{s}_vad"], atol=1e-4)
        np.testing.assert_allclose(out.cpu().numpy(), g[f"{s}_out"], atol=1e-4)


def _oracle_forward(m, w, n_fft, hop, padding=None):
    from oracle import spectral as osp
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    bufs = osp.stft_buffers(m.stft.max_frames, n_fft, hop)
    lm, ph = osp.stft_forward(w, bufs[0], n_fft, hop)
    enh, vad = osp.spectral_forward(lm, sd, padding=padding)
    return enh, vad, osp.model_forward(w, sd, bufs, n_fft, hop, padding=padding)


def _clip(B, L, seed):
    gen = torch.Generator().manual_seed(seed)
    env = 0.5 - 0.5 * torch.cos(2 * np.pi * 4.0 * torch.arange(L) / 16000.0)
    return (0.05 * torch.randn(B, 1, 1, L, generator=gen) * env + 0.02 * torch.randn(B, 1, 1, L, generator=gen)).clamp(-1, 1)


@pytest.mark.parametrize("tag,n_fft,hop", [("A", 320, 160), ("B", 512, 128)])
def test_forward_full_clips_vs_oracle(dev, golden_dir, tag, n_fft, hop):
    """The whole inference forward on 10 s clips (161x1001 -> W' 1009, 704 tokens; 257x1251 -> W' 1265, 1360
    tokens) against the CPU oracle: 1e-4 relative on the enhanced log-magnitudes, 1e-4 absolute on the waveform."""
    m = build(n_fft, hop, dev, golden_dir, max_frames=1300).eval()
    w = _clip(2, 160000, 7)
    enh_ref, vad_ref, out_ref = _oracle_forward(m, w, n_fft, hop)
    out = m(w.to(dev))
    lm, _ = m.stft.forward(w.to(dev))
    enh, vad = m._forward(lm)
    assert [m.model.H_padding, m.model.W_padding] == [0, 8 if tag == "A" else 14]
    assert rel(enh.cpu().numpy(), enh_ref.numpy()) < 1e-4
    np.testing.assert_allclose(vad.cpu().numpy(), vad_ref.numpy(), atol=1e-4)
    assert out.shape == out_ref.shape
    np.testing.assert_allclose(out.cpu().numpy(), out_ref.numpy(), atol=1e-4)


def test_padding_is_kept_and_skips_are_cropped(dev, golden_dir):
    """The reference searches the padding on the first call and keeps it for the model's lifetime; later inputs of
    other lengths then crop the skip tensors from the end (train/models.py:608).  Same model, three lengths, no
    reset in between, against the oracle run with the first call's padding."""
    m = build(320, 160, dev, golden_dir, max_frames=1300).eval()
    for L in (8000, 16000, 160000, 8000):
        w = _clip(2, L, L)
        out = m(w.to(dev))
        assert [m.model.H_padding, m.model.W_padding] == [0, 14]
        enh_ref, vad_ref, out_ref = _oracle_forward(m, w, 320, 160, padding=(0, 14))
        lm, _ = m.stft.forward(w.to(dev))
        enh, vad = m._forward(lm)
        assert rel(enh.cpu().numpy(), enh_ref.numpy()) < 1e-4, L
        np.testing.assert_allclose(vad.cpu().numpy(), vad_ref.numpy(), atol=1e-4)
        np.testing.assert_allclose(out.cpu().numpy(), out_ref.numpy(), atol=1e-4)
    # an input the kept padding cannot serve fails like the reference does (output smaller than the input)
    m2 = build(320, 160, dev, golden_dir, max_frames=1300).eval()
    m2(_clip(1, 160000, 1).to(dev))           # padding searched for 1001 frames: W_padding = 8
    with pytest.raises(RuntimeError):
        m2(_clip(1, 8000, 2).to(dev))


def test_graphed_forward_matches_eager(dev, golden_dir):
    """A HIP-graph replay of the eval forward is bit-identical to the eager launch sequence, on new input too."""
    g = np.load(os.path.join(golden_dir, "g2_eval_forward_A.npz"))
    m = build(320, 160, dev, golden_dir).eval()
    w = torch.from_numpy(g["seg_wave"]).to(dev)
    run = m.graphed_forward(w)
    for x in (w, 0.5 * w.flip(-1).contiguous()):
        ref = m(x).clone()
        got = run(x).clone()
        assert torch.equal(ref, got)
    np.testing.assert_allclose(run(w).cpu().numpy(), g["seg_out"], atol=1e-4)
    with pytest.raises(RuntimeError):
        m.train().graphed_forward(w)


def test_graphed_forward_survives_other_geometries(dev, golden_dir):
    """The captured graph addresses the plan's buffers directly: running more geometries than the plan cache holds,
    eagerly, must neither evict that plan nor let its memory be reused."""
    g = np.load(os.path.join(golden_dir, "g2_eval_forward_A.npz"))
    m = build(320, 160, dev, golden_dir).eval()
    w = torch.from_numpy(g["seg_wave"]).to(dev)
    run = m.graphed_forward(w)
    want = m(w).clone()
    for B in (1, 3, 4, 5, 6, 7):               # six other plans: more than the cache keeps
        m(torch.randn(B, 1, 1, 8000, device=dev).clamp(-1, 1) * 0.1)
        junk = torch.randn(1 << 22, device=dev)   # would land in any freed activation buffer
        del junk
    assert torch.equal(run(w), want)
    with pytest.raises(RuntimeError, match="graph"):
        m.model.reset_padding()
    del run
    m.model.reset_padding()                     # allowed again once the graph's owner is gone
    # the explicit dropout mask is validated and serves exactly one training step
    m.train()
    m.attn_keep_mask = torch.ones(2, 6, 55, 54, dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError, match="attn_keep_mask"):
        m.training_step((w, w, ["a", "b"]), 0)
    assert m.attn_keep_mask is None
    m.eval()
    with pytest.raises(RuntimeError, match="train mode"):
        m._grad_target = m._work_grads
        m._native_step(w, w, None, want_grad=True)


def _keep(g, which, dev):
    shape = tuple(g["attn_keep_shape"])
    bits = np.unpackbits(g[which])[: int(np.prod(shape))].reshape(shape)
    return torch.from_numpy(bits.astype(np.uint8)).to(dev).contiguous()


FIXTURES = ["g3t_train_step.npz", "g3_train_step.npz"]


def _load_train_fixture(golden_dir, fixture, dev, n_fft=320, hop=160):
    g = np.load(os.path.join(golden_dir, fixture))
    m = build(n_fft, hop, dev, golden_dir)
    tame = "last_weight" in g.files
    if tame:
        with torch.no_grad():
            m.model.decoder[-1][0].weight.copy_(torch.from_numpy(g["last_weight"]))
            m.model.decoder[-1][0].bias.copy_(torch.from_numpy(g["last_bias"]))
    return g, m.train(), tame


def _check_grads(named_grads, g, tame, tol=2e-3):
    """`tame` fixture (shrunken last layer, clamp of models.py:763 rarely active): every gradient within `tol`
    of the reference's, relative to the largest entry of that parameter.  Untamed fixture: half of the enhanced
    log-magnitudes sit on the clamp and the one element within 2e-6 of it (clip 3, bin 73, frame 36, which also
    carries a large waveform-loss gradient) tips over with the summation order of the BatchNorm statistics,
    moving every gradient by a few per cent (up to ~17 %); there no parameter may be off by more than 30 %,
    and the tight comparison is the `tame` fixture's."""
    errs = {}
    for k, gr in named_grads:
        ref = g["grad/" + k]
        # analytically-zero gradients (biases feeding a train-mode BatchNorm) are rounding noise on both sides
        scale = max(np.abs(ref).max(), 1e-4)
        errs[k] = float(np.abs(gr.detach().cpu().numpy() - ref).max() / scale)
    worst = max(errs.items(), key=lambda kv: kv[1])
    if tame:
        assert worst[1] < tol, f"gradient mismatch {worst}"
    else:
        assert worst[1] < 0.3, f"gradient mismatch {worst}"
    return worst[1] < tol


@pytest.mark.parametrize("fixture", FIXTURES)
def test_train_step_golden_fused(dev, golden_dir, fixture):
    """Fused path: forward + loss + backward + AdamW without autograd, two consecutive steps."""
    g, m, tame = _load_train_fixture(golden_dir, fixture, dev)
    opt = m.configure_optimizers()
    noisy, clean = torch.from_numpy(g["noisy"]).to(dev), torch.from_numpy(g["clean"]).to(dev)
    m.attn_keep_mask = _keep(g, "attn_keep1", dev)
    out4 = m.train_step_fused(noisy, clean, opt).cpu().numpy()
    # per-step loss in fp32: 1e-4 relative
    for i, k in enumerate(("loss", "vad_loss", "waveform_loss", "spectral_loss")):
        np.testing.assert_allclose(out4[i], g[f"log/train_{k}"], rtol=1e-4, err_msg=k)
    grads_tight = _check_grads([(k, p.grad) for k, p in m.named_parameters()], g, tame)
    for k, b in m.named_buffers():
        if "running" in k:
            np.testing.assert_allclose(b.cpu().numpy(), g["buf/" + k], rtol=1e-4, atol=1e-6, err_msg=k)
        elif "num_batches" in k:
            assert int(b.item()) == int(g["buf/" + k])
    for k, p in m.named_parameters():
        solid = np.abs(g["grad/" + k]) > 1e-5
        ref = g["param_after/" + k]
        got = p.detach().cpu().numpy()
        if grads_tight:     # Adam's first update is lr*sign(g): only meaningful where the gradients agree
            np.testing.assert_allclose(got[solid], ref[solid], atol=5e-6, err_msg=k)
        assert np.abs(got - ref).max() <= 2.1e-4, k
    m.attn_keep_mask = _keep(g, "attn_keep2", dev)
    out4 = m.train_step_fused(noisy, clean, opt).cpu().numpy()
    np.testing.assert_allclose(out4[0], g["loss2"], rtol=2e-4)


@pytest.mark.parametrize("fixture", FIXTURES)
def test_train_step_golden_autograd(dev, golden_dir, fixture):
    """Drop-in path: training_step(...) returns a loss whose .backward() fills p.grad; torch.optim.AdamW steps."""
    g, m, tame = _load_train_fixture(golden_dir, fixture, dev)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    noisy, clean = torch.from_numpy(g["noisy"]).to(dev), torch.from_numpy(g["clean"]).to(dev)
    m.attn_keep_mask = _keep(g, "attn_keep1", dev)
    opt.zero_grad()
    loss = m.training_step((noisy, clean, ["a"] * noisy.shape[0]), 0)
    assert loss.dim() == 0 and loss.requires_grad
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    assert set(m.logged_metrics) == {"train_loss", "train_vad_loss", "train_waveform_loss", "train_spectral_loss"}
    grads_tight = _check_grads([(k, p.grad) for k, p in m.named_parameters()], g, tame)
    opt.step()
    for k, p in m.named_parameters():
        solid = np.abs(g["grad/" + k]) > 1e-5
        if grads_tight:
            np.testing.assert_allclose(p.detach().cpu().numpy()[solid], g["param_after/" + k][solid], atol=5e-6, err_msg=k)
    # validation: eval-mode BatchNorm, no dropout, no gradient
    m.eval()
    with torch.no_grad():
        vloss = m.validation_step((noisy, clean, ["a"] * noisy.shape[0]), 0)
    assert not vloss.requires_grad and np.isfinite(vloss.item())
    assert "val_loss" in m.logged_metrics


def _oracle_grads(sd, noisy, clean, keep, dtype, padding=None, hop=160):
    from oracle import spectral as osp
    cast = (lambda v: v.to(dtype) if v.is_floating_point() else v)
    sdc = {k: cast(v) for k, v in sd.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sdc.items() if v.is_floating_point() and
              not ("running" in k or k.startswith("stft."))}
    full = dict(sdc)
    full.update(params)
    bufs = tuple(b.to(dtype) for b in osp.stft_buffers(1200, 320, hop))
    loss, _ = osp.shared_step(noisy.to(dtype), clean.to(dtype), full, bufs, 320, hop, training=True, new_stats={},
                              attn_keep=keep, padding=padding)
    loss.backward()
    return loss.item(), {k: v.grad.double().numpy() for k, v in params.items()}


def test_train_step_through_cropped_skips(dev, golden_dir):
    """The reference searches its padding on the first forward and keeps it (models.py:585-586, 628-639).  With the
    reference's hop of 160 every padding that lets an 8000-sample segment through (14 or 15 columns) leaves the skips
    uncropped; with hop 80 a model whose first input had 99 frames (W_padding 14) trains on segments of 101 frames whose
    full-resolution encoder outputs (58 columns) are one column wider than the decoder tensors coming back up (57), so
    `x + skip[..., -h:, -w:]` (models.py:608) crops and the gradient of each cropped skip is zero on the cut column."""
    m = build(320, 80, dev, golden_dir).eval()
    with torch.no_grad():
        m.model.decoder[-1][0].weight.mul_(0.02)
        m.model.decoder[-1][0].bias.zero_()
        m(torch.zeros(1, 1, 1, 98 * 80, device=dev))
    padding = (m.model.H_padding, m.model.W_padding)
    assert padding == (0, 14)
    m.train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    B = 5
    gen = torch.Generator().manual_seed(77)
    clean = (torch.randn(B, 1, 1, 8000, generator=gen) * 0.05).clamp(-1, 1)
    noisy = (clean + 0.03 * torch.randn(B, 1, 1, 8000, generator=gen)).clamp(-1, 1)
    n_tok = 11 * 8                     # 161 x (101 + 14) -> 81x58 -> 41x29 -> 21x15 -> 11x8
    keep = torch.rand(B, 6, n_tok, n_tok, generator=gen) > 0.1
    loss32, g32 = _oracle_grads(sd, noisy, clean, keep, torch.float32, padding, hop=80)
    loss64, g64 = _oracle_grads(sd, noisy, clean, keep, torch.float64, padding, hop=80)
    m.attn_keep_mask = keep.to(torch.uint8).to(dev).contiguous()
    out = m.training_step((noisy.to(dev), clean.to(dev), [""] * B), 0)
    out.backward()
    plan = m.model.plan(B, 161, 101, dev)
    assert sorted(plan.crops) == [3, 4] and all(ox == 1 and oy == 0 for _, oy, ox in plan.crops.values())
    np.testing.assert_allclose(out.item(), loss64, rtol=1e-4)
    worst = ("", 0.0, 0.0)
    for k, p in m.named_parameters():
        ref = g64[k]
        scale = max(np.abs(ref).max(), 1e-4)
        e_gpu = np.abs(p.grad.cpu().double().numpy() - ref).max() / scale
        e_cpu = np.abs(g32[k] - ref).max() / scale
        allowed = max(2e-3, 2.0 * e_cpu)
        if e_gpu / allowed > worst[1]:
            worst = (k, e_gpu / allowed, e_gpu)
    assert worst[1] < 1.0, worst


@pytest.mark.parametrize("B,tame", [(1, True), (7, True), (33, True), (7, False)])
def test_train_step_vs_oracle(dev, golden_dir, B, tame):
    """Random batches (odd sizes, silent clips) against the CPU oracle with an explicit dropout mask.

    With the fixture's random weights the enhanced waveform saturates the [-1,1] clamp, and the loss gradient
    (signs of an L1, clamp masks) becomes so ill-conditioned that the fp32 oracle itself is ~4 % away from a
    float64 evaluation.  `tame` shrinks the last layer so the problem is well conditioned and a tight fp32
    tolerance is meaningful; the untamed case is judged relative to the fp32 oracle's own distance to float64."""
    m = build(320, 160, dev, golden_dir).train()
    if tame:
        with torch.no_grad():
            m.model.decoder[-1][0].weight.mul_(0.02)
            m.model.decoder[-1][0].bias.zero_()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(100 + B)
    clean = (torch.randn(B, 1, 1, 8000, generator=gen) * 0.05).clamp(-1, 1)
    if B > 2:
        clean[1] = 0
    noisy = (clean + 0.03 * torch.randn(B, 1, 1, 8000, generator=gen)).clamp(-1, 1)
    keep = torch.rand(B, 6, 55, 55, generator=gen) > 0.1
    loss32, g32 = _oracle_grads(sd, noisy, clean, keep, torch.float32)
    loss64, g64 = _oracle_grads(sd, noisy, clean, keep, torch.float64)
    m.attn_keep_mask = keep.to(torch.uint8).to(dev).contiguous()
    out = m.training_step((noisy.to(dev), clean.to(dev), [""] * B), 0)
    out.backward()
    np.testing.assert_allclose(out.item(), loss64, rtol=1e-4)
    worst = ("", 0.0, 0.0)
    for k, p in m.named_parameters():
        ref = g64[k]
        scale = max(np.abs(ref).max(), 1e-4)
        e_gpu = np.abs(p.grad.cpu().double().numpy() - ref).max() / scale
        e_cpu = np.abs(g32[k] - ref).max() / scale
        # B = 1 puts only 55 samples behind the bottleneck BatchNorm statistics, which amplifies fp32 rounding
        allowed = max(1e-2 if B == 1 else 2e-3, 2.0 * e_cpu)
        if e_gpu / allowed > worst[1]:
            worst = (k, e_gpu / allowed, e_gpu)
    assert worst[1] < 1.0, worst


def test_dropout_rng_statistics(dev, golden_dir):
    """Without an explicit mask the attention dropout comes from the in-kernel counter RNG: the step must
    still be finite, differ from step to step, and drop about 10 % (checked through determinism of eval)."""
    m = build(320, 160, dev, golden_dir).train()
    from noise_amd.data import synth_batch
    noisy, clean, names = synth_batch(8, 8000, dev, seed=3)
    with torch.no_grad():
        a = m.training_step((noisy, clean, names), 0).item()
        b = m.training_step((noisy, clean, names), 1).item()
    assert np.isfinite(a) and np.isfinite(b) and a != b
    m.eval()
    with torch.no_grad():
        c = m.validation_step((noisy, clean, names), 0).item()
        d = m.validation_step((noisy, clean, names), 1).item()
    assert c == d


def test_mobilenet_forward_interface(dev, golden_dir):
    """VADMobileNetV2._forward(x) keeps the reference contract: normalised (B,1,H,W) -> filters, vad."""
    from oracle import spectral as osp
    m = build(320, 160, dev, golden_dir).eval()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x = torch.rand(2, 1, 161, 51) * 2 - 1
    filt_ref, vad_ref = osp.denoiser(x, sd)
    filt, vad = m.model._forward(x.to(dev))
    assert rel(filt.cpu().numpy(), filt_ref.numpy()) < 1e-4
    np.testing.assert_allclose(vad.cpu().numpy(), vad_ref.numpy(), atol=1e-4)
