"""GPU parity: float64 SNR mixer / activity detector against the golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_mixer_golden(dev, golden_dir):
    from noise_amd import mixer
    g = np.load(os.path.join(golden_dir, "g4_mixer.npz"))
    for i in range(3):
        c, n, y, lv = mixer.mix_batch(torch.from_numpy(g[f"case{i}_clean_in"])[None].to(dev),
                                      torch.from_numpy(g[f"case{i}_noise_in"])[None].to(dev),
                                      torch.tensor([int(g[f"case{i}_snr"])], device=dev),
                                      torch.tensor([int(g[f"case{i}_level_draw"])], device=dev))
        # float64 arithmetic in the reference's operation order; only reduction order differs
        for got, key in ((c, "clean"), (n, "noise"), (y, "noisy")):
            np.testing.assert_allclose(got[0].cpu().numpy(), g[f"case{i}_{key}"], rtol=1e-12, atol=1e-15)
        assert int(lv.item()) == int(g[f"case{i}_level"])
        act = mixer.activity_batch(torch.from_numpy(g[f"case{i}_clean_in"])[None].to(dev))
        assert abs(float(act.item()) - float(g[f"case{i}_activity_clean"])) < 1e-12
        act = mixer.activity_batch(torch.from_numpy(g[f"case{i}_noisy"])[None].to(dev))
        assert abs(float(act.item()) - float(g[f"case{i}_activity_noisy"])) < 1e-12


def test_mixer_batch_vs_oracle(dev):
    from noise_amd import mixer
    from oracle import mixer as omix
    rng = np.random.RandomState(3)
    U, L = 7, 16000 * 3 + 137          # ragged last 100 ms / 50 ms windows
    clean = rng.randn(U, L) * 0.1 * rng.rand(U, 1)
    noise = rng.randn(U, L) * 0.05
    clean[3] *= 1 + 300 * (rng.rand(L) > 0.9995)     # forces the de-clipping branch
    snr = rng.randint(-5, 20, size=U)
    level = rng.randint(-35, -15, size=U)
    level[3] = -16
    c, n, y, lv = mixer.mix_batch(torch.from_numpy(clean).to(dev), torch.from_numpy(noise).to(dev),
                                  torch.from_numpy(snr).to(dev), torch.from_numpy(level).to(dev))
    act = mixer.activity_batch(y)
    took_branch = False
    for u in range(U):
        rc, rn, ry, rl = omix.segmental_snr_mixer(clean[u], noise[u], int(snr[u]), int(level[u]))
        np.testing.assert_allclose(c[u].cpu().numpy(), rc, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(n[u].cpu().numpy(), rn, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(y[u].cpu().numpy(), ry, rtol=1e-12, atol=1e-15)
        assert int(lv[u].item()) == rl
        took_branch |= rl != int(level[u])
        assert abs(float(act[u].item()) - omix.activitydetector(ry)) < 1e-12
        # properties: mix is the sum of its parts; nothing exceeds the clipping threshold after the fix-up
        np.testing.assert_allclose((c[u] + n[u]).cpu().numpy(), y[u].cpu().numpy(), rtol=0, atol=1e-15)
    assert took_branch
    assert float(y[3].abs().max().item()) <= 0.99
    # the peaks and the float32 mix written by the mixing pass itself equal what separate passes would compute
    c2, n2, y2, lv2, peaks, y32 = mixer.mix_batch(torch.from_numpy(clean).to(dev), torch.from_numpy(noise).to(dev),
                                                  torch.from_numpy(snr).to(dev), torch.from_numpy(level).to(dev),
                                                  with_peaks=True)
    assert torch.equal(c2, c) and torch.equal(n2, n) and torch.equal(y2, y) and torch.equal(lv2, lv)
    want = torch.stack([c.abs().amax(1), n.abs().amax(1), y.abs().amax(1)], 1)
    assert torch.equal(peaks[:, :3], want)
    np.testing.assert_allclose(peaks[:, 3].cpu().numpy(), y.sum(1).cpu().numpy(), rtol=1e-11, atol=1e-12)
    assert torch.equal(y32, y.to(torch.float32))


def test_mixer_drop_in_signature(dev):
    from noise_amd import mixer
    from oracle import mixer as omix
    rng = np.random.RandomState(9)
    clean, noise = rng.randn(20000) * 0.1, rng.randn(16000) * 0.02      # unequal lengths: shorter one is zero-padded
    params = {"cfg": None, "target_level_lower": -35, "target_level_upper": -15}
    np.random.seed(11)
    c, n, y, lv = mixer.segmental_snr_mixer(params, clean, noise, 7)
    np.random.seed(11)
    level = np.random.randint(-35, -15)
    rc, rn, ry, rl = omix.segmental_snr_mixer(clean, noise, 7, level)
    assert len(y) == 20000 and lv == rl
    np.testing.assert_allclose(y, ry, rtol=1e-12, atol=1e-15)
    assert abs(mixer.activitydetector(y) - omix.activitydetector(ry)) < 1e-12
    assert mixer.mix_batch(torch.zeros(0, 10, dtype=torch.float64, device=dev), torch.zeros(0, 10, dtype=torch.float64, device=dev),
                           torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev))[0].shape == (0, 10)
