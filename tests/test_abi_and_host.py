"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares, and the
host-side logic (module tree, padding search, flat parameters, data contract) behaves like the reference."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYER_CONFIG = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [3, 24, 1, [3, 3], [2, 2]],
                [3, 32, 1, [3, 3], [2, 2]], [3, 64, 1, [3, 3], [2, 2]]]


def _model(n_fft=320, hop=160, max_frames=300):
    from noise_amd import VADSpectralV2
    return VADSpectralV2(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann", nhead=6, num_layers=1,
                         dropout=0.0, layer_config=[list(r) for r in LAYER_CONFIG], pad_value=-1, max_frames=max_frames)


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "noise_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(noise_[A-Za-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 35
    lib = ctypes.CDLL(os.path.join(ROOT, "noise_amd", "libnoise_hip.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.noise_version.restype = ctypes.c_int
    assert lib.noise_version() >= 100
    # every bound Python signature refers to a declared entry point
    from noise_amd import _lib, ops, engine, augment, losses  # noqa: F401
    assert set(_lib._SIGNATURES) <= set(names)


def test_bad_arguments_fail_before_any_launch():
    from noise_amd import _lib
    with pytest.raises(RuntimeError, match="n_fft"):
        _lib.call("noise_stft_fwd", None, None, None, None, None, 1, 8000, 300, 150, None)
    with pytest.raises(RuntimeError, match="null"):
        _lib.call("noise_stft_fwd", None, None, None, None, None, 1, 8000, 320, 160, None)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        _lib.call("noise_dw3x3_fwd", None, None, 0, None, None, 0, None, None, None, None, 0, 1, 4, 4, 6, 1, 1, None)


def test_cpu_tensors_are_rejected():
    m = _model()
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 1, 1, 8000))
    with pytest.raises(RuntimeError, match="GPU"):
        m.stft.forward(torch.zeros(1, 1, 1, 8000))
    from noise_amd import mixer
    with pytest.raises(RuntimeError, match="GPU"):
        mixer.mix_batch(torch.zeros(1, 10, dtype=torch.float64), torch.zeros(1, 10, dtype=torch.float64),
                        torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32))


def test_state_dict_layout_matches_reference(golden_dir):
    m = _model(max_frames=1200)
    want = [l.split(" ", 1) for l in open(os.path.join(golden_dir, "g2_state_dict_keys.txt")).read().splitlines()]
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _ in want]
    for k, rest in want:
        shape = eval(rest.rsplit(" ", 1)[0])
        assert tuple(sd[k].shape) == shape, k
    assert sum(p.numel() for p in m.parameters()) == 76146
    assert m.hparams.n_fft == 320 and m.hparams.pad_value == -1
    assert m.SUPPORTED_SAMPLING_RATE == 16000


def test_default_init_matches_reference_rng_order(golden_dir):
    """Built under the same seed, conv / linear weights equal the reference's (its custom init loop is a no-op)."""
    torch.manual_seed(0)
    m = _model(max_frames=1200)
    g = np.load(os.path.join(golden_dir, "g2_state_dict.npz"))
    sd = m.state_dict()
    same = [k for k in g.files if sd[k].dim() > 1 and k != "model.vad.weight" and np.array_equal(g[k], sd[k].numpy())]
    assert len(same) == len([k for k in g.files if sd[k].dim() > 1]) - 1


def test_stft_buffers_match_reference(golden_dir):
    from noise_amd import STFT
    g = np.load(os.path.join(golden_dir, "g1_stft.npz"))
    for tag, n_fft, hop in (("A", 320, 160), ("B", 512, 128)):
        st = STFT(400, n_fft, hop, n_fft, "hann")
        rows = g[f"{tag}_basis_rows"]
        np.testing.assert_allclose(st.forward_basis[:, 0].numpy()[rows], g[f"{tag}_forward_basis_rows"], atol=1e-7)
        np.testing.assert_allclose(st.inverse_basis[:, 0].numpy()[rows], g[f"{tag}_inverse_basis_rows"], atol=1e-7)
        assert np.array_equal(st.window_sum.numpy()[: 4 * n_fft], g[f"{tag}_window_sum_head"])
        assert np.array_equal(st.window_sum.numpy()[-2 * n_fft:], g[f"{tag}_window_sum_tail"])
        assert st.window_sum.numel() == n_fft + hop * 399


def test_padding_search_closed_form():
    from noise_amd import VADMobileNetV2
    for (H, W), want in {(161, 51): (0, 14), (161, 63): (0, 2), (161, 1001): (0, 8), (257, 1251): (0, 14),
                         (161, 101): (0, 12), (160, 49): (1, 0)}.items():
        net = VADMobileNetV2(6, 1, 0.0, layer_config=[list(r) for r in LAYER_CONFIG], pad_value=-1)
        net._search_padding(H, W)
        assert (net.H_padding, net.W_padding) == want, (H, W)
        assert net.found and net.searching      # like the reference, the search runs once per model


def test_flat_parameters_are_views():
    m = _model()
    params = m._ensure_flat()
    flat = m._flat_params
    assert flat.numel() == 76146
    o = 0
    for p in params:
        assert p.data_ptr() == flat.data_ptr() + 4 * o
        o += p.numel()
    before = m.model.vad.bias.item()
    flat[-1] += 1.0
    assert m.model.vad.bias.item() == pytest.approx(before + 1.0, abs=1e-6)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert m._ensure_flat()[0].data_ptr() == flat.data_ptr()      # loading a checkpoint keeps the views


def test_unsupported_configurations_say_so():
    from noise_amd import VADMobileNetV2, VADSpectralV2
    # two encoder layers / a residual block: the containers build (the shape-generic forward serves them), the tuned
    # engine and the spectral training model say what they do not cover
    for layers, lc in [(2, [list(r) for r in LAYER_CONFIG]), (1, [[1, 32, 1, [3, 3], [2, 2]], [1, 32, 1, [3, 3], [1, 1]]])]:
        m = VADMobileNetV2(6, layers, 0.0, layer_config=lc)
        assert m.untuned
        with pytest.raises(NotImplementedError, match="tuned kernel schedule"):
            m.plan(1, 161, 51, "cpu")
        with pytest.raises(NotImplementedError, match="spectral training engine"):
            VADSpectralV2(n_fft=320, hop_length=160, win_length=320, window="hann", nhead=6, num_layers=layers, dropout=0.0,
                          layer_config=lc, pad_value=-1, max_frames=400)
    assert VADMobileNetV2(6, 1, 0.0, layer_config=[list(r) for r in LAYER_CONFIG]).untuned is None
    with pytest.raises(NotImplementedError):
        VADMobileNetV2(6, 1, 0.0, layer_config=[[1, 32, 1, [5, 5], [2, 2]], [1, 16, 1, [3, 3], [1, 1]]])
    with pytest.raises(NotImplementedError):
        VADMobileNetV2(6, 1, 0.1, layer_config=[list(r) for r in LAYER_CONFIG])
    with pytest.raises(ValueError):
        VADMobileNetV2(6, 1, 0.0, layer_config=[])


def test_synthetic_datamodule_contract():
    from noise_amd import VADNoiseDataModule
    dm = VADNoiseDataModule(data_dir=None, sampling_rate=16000, length=8000, num_samples=64, num_workers=4,
                            batch_size=16, p=0.3, device="cpu")
    dm.setup("fit")
    batches = list(dm.train_dataloader())
    assert len(batches) == 4
    noisy, clean, names = batches[0]
    assert noisy.shape == clean.shape == (16, 1, 1, 8000) and noisy.dtype == torch.float32
    assert float(noisy.abs().max()) <= 1.0 and len(names) == 16 and isinstance(names[0], str)
    labels = ((clean.abs() > 0).sum(-1) >= 160).float()
    assert 0 < labels.mean() <= 1
    again = next(iter(VADNoiseDataModule(length=8000, num_samples=64, batch_size=16, device="cpu").train_dataloader()))
    assert torch.equal(again[0], noisy)
    other_rank = next(iter(VADNoiseDataModule(length=8000, num_samples=64, batch_size=16, device="cpu", rank=1,
                                              world_size=2).train_dataloader()))
    assert not torch.equal(other_rank[0], noisy)


def test_checkpoint_round_trip(tmp_path):
    """`load_from_checkpoint` (train/export.py:8) reads the "hyper_parameters" and "state_dict" entries of a Lightning
    checkpoint with a loader that executes nothing from the file."""
    import torch
    from noise_amd import VADSpectralV2
    lc = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [3, 24, 1, [3, 3], [2, 2]], [3, 32, 1, [3, 3], [2, 2]],
          [3, 64, 1, [3, 3], [2, 2]]]
    torch.manual_seed(3)
    m = VADSpectralV2(n_fft=320, hop_length=160, win_length=320, window="hann", nhead=6, num_layers=1, dropout=0.0,
                      layer_config=lc, pad_value=-1, max_frames=40)
    if not hasattr(m, "save_checkpoint"):
        pytest.skip("Lightning is installed: its own checkpoint I/O applies")
    path = str(tmp_path / "spectralv2.ckpt")
    m.save_checkpoint(path, epoch=192)
    torch.manual_seed(99)
    m2 = VADSpectralV2.load_from_checkpoint(path)
    assert dict(m2.hparams) == dict(m.hparams) and m2.hparams.max_frames == 40
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert list(sd) == list(sd2) and len(sd) == 159
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)
    m3 = VADSpectralV2.load_from_checkpoint(path, max_frames=40, pad_value=-1)
    assert m3.hparams.pad_value == -1


def test_product_path_has_no_cpu_stand_in(tmp_path):
    """The CPU oracle is test infrastructure: nothing under noise_amd/ may name it, bench.py may only import it inside
    its CPU-baseline legs, and importing the package without the built library must raise instead of degrading."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "noise_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{f} imports the oracle"
    bench = open(os.path.join(root, "bench.py"), encoding="utf-8").read()
    imports = [m.start() for m in re.finditer(r"^\s*(from|import)\s+oracle\b", bench, re.M)]
    legs = []
    for name in ("def cpu_baseline", "def other_configs_cpu"):       # the two CPU-baseline legs of the bench line
        start = bench.index(name)
        legs.append((start, bench.index("\ndef ", start + 1)))
    assert imports and all(any(a < i < b for a, b in legs) for i in imports), \
        "bench.py may use the oracle in its CPU-baseline legs only"
    # a missing library is an ImportError that names the file, not a silent fallback
    env = dict(os.environ, NOISE_HIP_LIB=str(tmp_path / "absent.so"), PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", "import noise_amd"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "absent.so is missing" in r.stderr and "no CPU fallback" in r.stderr
