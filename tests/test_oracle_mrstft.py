"""The numpy oracle of the waveform models' loss against fixtures written by the reference itself
(tests/golden/make_golden.py, G5: Base.stft_mag / Base.multi_resolution_stft_loss called on the reference classes)."""
import os

import numpy as np
import pytest

from oracle import mrstft as om

CFG = ([512, 1024, 2048], [50, 120, 240], [240, 600, 1200])


@pytest.fixture(scope="module")
def g5(golden_dir):
    return np.load(os.path.join(golden_dir, "g5_mrstft.npz"))


def test_magnitudes(g5):
    x = g5["enhanced"].reshape(3, -1)
    for i, (n, h, w) in enumerate(zip(*CFG)):
        mag = om.stft_mag(x, n, h, w)
        assert list(mag.shape) == list(g5[f"mag_{i}_shape"])
        np.testing.assert_allclose(mag[:, g5[f"mag_{i}_frames"]], g5[f"mag_{i}"], rtol=2e-4, atol=2e-6)
    assert (g5["mag_0"] == np.float32(om.CLAMP)).any(), "the fixture exercises the clamp"


def test_loss_and_gradient(g5):
    x, y = g5["enhanced"], g5["clean"]
    loss, grad, parts = om.multi_resolution_stft_loss(x, y, *CFG, want_grad=True)
    np.testing.assert_allclose(loss, g5["loss"], rtol=2e-6)
    # the reference transforms in fp32; a bin where |X| and |Y| agree to rounding can take the other sign there
    np.testing.assert_allclose(grad, g5["grad"], atol=2e-5 * np.abs(g5["grad"]).max() + 1e-5)
    for i in range(3):
        np.testing.assert_allclose(parts[i], g5[f"loss_{i}"], rtol=2e-6)
        _, gi = om.resolution_loss_grad(x.reshape(3, -1), y.reshape(3, -1), CFG[0][i], CFG[1][i], CFG[2][i])
        np.testing.assert_allclose(gi, g5[f"grad_{i}"].reshape(3, -1), atol=1e-5)


def test_odd_length(g5):
    loss, grad, _ = om.multi_resolution_stft_loss(g5["odd_enhanced"], g5["odd_clean"], *CFG, want_grad=True)
    np.testing.assert_allclose(loss, g5["odd_loss"], rtol=2e-6)
    np.testing.assert_allclose(grad, g5["odd_grad"], atol=5e-5)


def test_total_loss(g5):
    total, parts, gx, _ = om.waveform_loss(g5["enhanced"], g5["clean"], g5["total_vad_logit"], float(g5["total_lambda"]), *CFG)
    np.testing.assert_allclose(total, g5["total_loss"], rtol=2e-6)
    np.testing.assert_allclose(parts["l1"], g5["total_l1"], rtol=1e-5)
    np.testing.assert_allclose(parts["vad"], g5["total_vad"], rtol=1e-6)
    np.testing.assert_allclose(gx, g5["total_grad"], atol=1e-5)


def test_gradient_is_the_derivative():
    """Finite differences of the oracle's own loss (double precision) against its analytic gradient."""
    rng = np.random.RandomState(0)
    x = rng.randn(1, 1500) * 0.1
    y = rng.randn(1, 1500) * 0.1
    _, g = om.resolution_loss_grad(x, y, 512, 50, 240)
    for s in (0, 3, 255, 256, 700, 1243, 1244, 1499):
        e = np.zeros_like(x)
        e[0, s] = 1e-6
        lp, _ = om.resolution_loss_grad(x + e, y, 512, 50, 240)
        lm, _ = om.resolution_loss_grad(x - e, y, 512, 50, 240)
        np.testing.assert_allclose((lp - lm) / 2e-6, g[0, s], rtol=1e-4, atol=1e-9)
