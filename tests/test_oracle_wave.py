"""The CPU oracle of the waveform-domain U-Net forward (oracle/wave_unet.py) against fixture G9, which the reference
itself wrote (tests/golden/make_golden_wave.py), and the host-side container tree against the reference's state_dict."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_wave as MG  # noqa: E402  (configurations, perturb(); main() is never called here)

from noise_amd.models import VADMobileNetV2, VADSpectralV2  # noqa: E402
from oracle import wave_unet as owu  # noqa: E402

G9 = np.load(os.path.join(HERE, "golden", "g9_wave.npz"))


def build(name, linear_gain=8.0):
    cfg, seed, _, _ = MG.CASES[name]
    torch.manual_seed(seed)
    m = VADMobileNetV2(**MG.kwargs(cfg))
    MG.perturb(m, seed, linear_gain)
    with torch.no_grad():
        m.vad.bias.copy_(torch.from_numpy(G9[f"{name}_vad_bias"]))
    return m


def keep_masks(name, m):
    out = []
    for i in range(len(m.bottleneck_attention.layer_stack)):
        shape = tuple(int(v) for v in G9[f"{name}_keep{i}_shape"])
        bits = np.unpackbits(G9[f"{name}_keep{i}"])[:int(np.prod(shape))]
        out.append(torch.from_numpy(bits.reshape(shape).astype(np.uint8)))
    return out


def test_state_dict_matches_the_reference():
    torch.manual_seed(31)
    m = VADMobileNetV2(**MG.kwargs(MG.YAML))
    keys = [f"{k} {tuple(v.shape)}" for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, "golden", "g9_wave_keys.txt")) as f:
        assert keys == f.read().split("\n")[:-1]
    sums = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in m.state_dict().values()])
    np.testing.assert_allclose(sums, G9["yaml_checksums"], rtol=1e-12, atol=1e-12)   # same constructor RNG order
    assert m.untuned and "residual" in m.untuned
    with pytest.raises(NotImplementedError):
        m.plan(1, 1, 8000, "cpu")
    with pytest.raises(NotImplementedError):     # the spectral training engine keeps refusing such a denoiser
        VADSpectralV2(n_fft=320, hop_length=160, win_length=320, window="hann", nhead=6, num_layers=2, dropout=0.0,
                      layer_config=MG.kwargs(MG.YAML)["layer_config"], pad_value=-1, max_frames=400)


@pytest.mark.parametrize("name", list(MG.CASES))
def test_oracle_eval_forward(name):
    m = build(name).eval()
    x1, x2 = torch.from_numpy(G9[f"{name}_x1"]), torch.from_numpy(G9[f"{name}_x2"])
    filt, vad = owu.forward(m, owu.normalize(x1)[0], False)
    assert (m.H_padding, m.W_padding) == tuple(G9[f"{name}_padding"])
    np.testing.assert_allclose(filt.numpy(), G9[f"{name}_eval_filters"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(vad.numpy(), G9[f"{name}_eval_vad"], rtol=0, atol=2e-6)
    # the gate sigmoid(1e4 * vad) magnifies the logit's rounding 2500-fold
    np.testing.assert_allclose(owu.full_forward(m, x1, False).numpy(), G9[f"{name}_eval_y1"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(owu.full_forward(m, x2, False).numpy(), G9[f"{name}_eval_y2"], rtol=0, atol=2e-3)


@pytest.mark.parametrize("name", list(MG.CASES))
def test_oracle_train_forward(name):
    m = build(name).eval()
    x1 = torch.from_numpy(G9[f"{name}_x1"])
    owu.forward(m, owu.normalize(x1)[0], False)          # padding search in eval mode, like the generator
    m.train()
    filt, vad = owu.forward(m, owu.normalize(x1)[0], True, keep_masks(name, m))
    np.testing.assert_allclose(filt.numpy(), G9[f"{name}_train_filters"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(vad.numpy(), G9[f"{name}_train_vad"], rtol=0, atol=5e-6)
    bufs = dict(m.named_buffers())
    for key in G9.files:
        if key.startswith(f"{name}_buf/"):
            np.testing.assert_allclose(bufs[key.split("/", 1)[1]].numpy(), G9[key], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["narrow", "square"])
def test_oracle_search_in_train_mode_moves_the_statistics(name):
    m = build(name).train()
    m.attn_dropout = 0.0
    x1 = torch.from_numpy(G9[f"{name}_x1"])
    filt, vad = owu.forward(m, owu.normalize(x1)[0], True)
    np.testing.assert_allclose(filt.numpy(), G9[f"{name}_fresh_train_filters"], rtol=0, atol=5e-5)
    bufs = dict(m.named_buffers())
    for key in G9.files:
        if key.startswith(f"{name}_fresh/"):
            np.testing.assert_allclose(bufs[key.split("/", 1)[1]].numpy(), G9[key], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["yaml", "narrow"])
def test_oracle_validation_step(name):
    """Base._shared_step through validation_step: total and the three logged parts."""
    cfg = MG.CASES[name][0]
    m = build(name).eval()
    x1, clean = torch.from_numpy(G9[f"{name}_x1"]), torch.from_numpy(G9[f"{name}_val_clean"])
    owu.forward(m, owu.normalize(x1)[0], False)           # the generator's first call: padding found on x1's shape
    total, parts = owu.shared_step(m, x1, clean, cfg.get("mr_stft_lambda", 0.5), [512, 1024, 2048], [50, 120, 240],
                                   [240, 600, 1200])
    want = G9[f"{name}_val_losses"]
    np.testing.assert_allclose([total, total, parts["l1"], parts["mrstft"], parts["vad"]], want, rtol=2e-5)


G10 = np.load(os.path.join(HERE, "golden", "g10_wave_train.npz"))


def test_oracle_train_step():
    """Base.training_step + backward on the narrow case: the oracle in fp32 against the reference's gradients, and a float64
    evaluation as the yardstick of what fp32 is worth on this step (batch statistics over two clips, ReLU6 masks)."""
    name = "narrow"
    cfg = MG.CASES[name][0]
    masks = []
    for i in range(2):
        shape = tuple(int(v) for v in G10[f"{name}_keep{i}_shape"])
        bits = np.unpackbits(G10[f"{name}_keep{i}"])[:int(np.prod(shape))]
        masks.append(torch.from_numpy(bits.reshape(shape).astype(np.uint8)))
    names, stats = list(G10[f"{name}_grad_names"]), G10[f"{name}_grad_stats"]
    floor = 1e-7 * float(stats[:, 2].max())
    worst = {}
    for dt in (torch.float32, torch.float64):
        m = build(name, linear_gain=1.0).eval()
        x1, clean = torch.from_numpy(G9[f"{name}_x1"]), torch.from_numpy(G9[f"{name}_val_clean"])
        owu.forward(m, owu.normalize(x1)[0], False)
        m.train()
        losses, grads = owu.train_step(m, x1, clean, masks, cfg.get("mr_stft_lambda", 0.5), [512, 1024, 2048], [50, 120, 240],
                                       [240, 600, 1200], dtype=dt)
        np.testing.assert_allclose(losses, G10[f"{name}_losses"], rtol=1e-4)
        assert list(grads) == names
        worst[dt] = max(max(float((grads[k].double() - torch.from_numpy(G10[f"{name}_grad/{k}"]).double()).norm()) - floor, 0.0)
                        / max(nrm, 1e-30) for k, (_, _, nrm) in zip(names, stats))
    assert worst[torch.float32] < 2e-2 and worst[torch.float64] < 4e-2, worst
