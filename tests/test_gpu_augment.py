"""Waveform augmentation kernels (csrc/augment.hip) through the C-ABI against the reference-written fixture
(tests/golden/g6_augment.npz) and the CPU restatement (oracle/augment.py).

Tolerance: the partitioned fp32 FFT convolution and scipy's single fp32 FFT both carry rounding of a few 1e-7 of the
largest output sample per butterfly level; 3e-5 of the largest |output| is asserted (measured ~2e-6), 1e-5 absolute
on the RMS-matched [-1, 1] results of the elementwise chain."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import augment as oa

G6 = os.path.join(os.path.dirname(__file__), "golden", "g6_augment.npz")


def _aug():
    from noise_amd import augment
    return augment


def _close(got, want, rel=3e-5):
    scale = float(np.max(np.abs(want))) or 1.0
    err = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))))
    assert err <= rel * scale, f"max |diff| {err:.3e} > {rel:.0e} * {scale:.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ir_short", "ir_room", "ir_hall"])
def test_reverb_vs_reference_fixture(name):
    g6 = np.load(G6)
    aug = _aug()
    speech, ir = g6["speech"], g6[name]
    _close(aug.add_pyreverb(speech, ir), g6[f"add_pyreverb_{name}"])
    got = aug.impulse_response(speech, ir)
    want = g6[f"impulse_response_{name}"]
    assert got.shape == want.shape and np.abs(got).max() <= 1.0
    _close(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("L,n_ir", [(700, 1), (1024, 1024), (1025, 1025), (2500, 3000), (5000, 2048), (40000, 9000),
                                    (4096, 1), (3073, 2047)])
def test_convolution_ragged_lengths(L, n_ir):
    aug = _aug()
    rng = np.random.RandomState(L + n_ir)
    x = rng.randn(3, L).astype(np.float32)
    irs = [rng.randn(n_ir).astype(np.float32) * np.exp(-np.arange(n_ir) / (0.3 * n_ir + 1)).astype(np.float32),
           np.array([0.0, 0.0, 1.0], dtype=np.float32)]
    pool = aug.ImpulseResponsePool(irs)
    idx = torch.tensor([0, 1, 0], dtype=torch.int32, device="cuda")
    y = aug.fir_convolve(torch.from_numpy(x).cuda(), idx, pool).cpu().numpy()
    for r in range(3):
        want = np.convolve(x[r].astype(np.float64), irs[int(idx[r])].astype(np.float64))[:L]
        _close(y[r], want)
    # a delayed unit impulse moves the row by two samples
    np.testing.assert_allclose(y[1][2:], x[1][:-2], atol=2e-6 * np.abs(x[1]).max() * 10)


def _pools(aug, N=5, L=40000, seed=7, paired=False):
    rng = np.random.RandomState(seed)
    env = 0.5 * (1 - np.cos(2 * np.pi * 4 * np.arange(L) / 16000.0))
    clean = (rng.randn(N, L) * 0.1 * env).astype(np.float32)
    noisy = (clean + rng.randn(N, L).astype(np.float32) * 0.02).astype(np.float32)
    noises = [(rng.randn(L + 7000) * 0.05).astype(np.float32), (rng.randn(3001) * 0.3).astype(np.float32),
              np.zeros(5000, dtype=np.float32)]
    irs = []
    for n, decay in ((300, 60.0), (4000, 700.0), (11000, 3000.0)):
        h = rng.randn(n) * np.exp(-np.arange(n) / decay)
        h[0] = 1.0
        irs.append(h.astype(np.float32))
    pools = aug.AugmentPools(clean, noises, irs, noisy=noisy if paired else None)
    return pools, noisy, clean, noises, irs


@pytest.mark.gpu
def test_vad_chain_vs_oracle():
    aug = _aug()
    pools, noisy, clean, noises, irs = _pools(aug)
    L, length = pools.clip_length, 8000
    t = aug.VADTrainAugment(pools, length, p=0.5, seed=3)
    noise_pool = np.concatenate(noises)
    off = pools.noise_offsets
    rows = [  # src shift gain snr noise_off noise_len ir cut invert
        (0, 0, 1.0, 10.0, off[0] + 100, L, -1, 0, 0),
        (1, 5000, 0.3, -5.0, off[0], L, 1, 12000, 1),
        (2, -7001, 1.7, 20.0, off[1], 3001, 0, L - length, 0),          # short noise repeats
        (3, 123, 0.9, 3.0, off[2], 5000, 2, 777, 1),                     # silent noise: clip unchanged
        (4, L, 1.0, 0.0, off[0] + 6999, L, -1, 16000, 0),                # shifted out entirely
        (0, -L, 1.0, 0.0, off[1] + 1000, 2001, 1, 31999, 1),
        (2, 1, 0.18, 12.5, 0, 0, 2, 5, 0),                               # no noise at all
        (1, -1, 1.0, 7.0, off[0] + 1, L, -1, 20000, 1),
    ]
    items = np.zeros(len(rows), dtype=aug.ITEM_DTYPE)
    for n, r in enumerate(rows):
        for k, v in zip(("src", "shift", "gain", "snr_db", "noise_off", "noise_len", "ir", "cut", "invert"), r):
            items[n][k] = v
    items = t.assign_rows(items)
    got_noisy, got_clean = t.apply(items)
    assert got_noisy.shape == (len(rows), 1, 1, length) and got_clean.shape == got_noisy.shape
    gn, gc = got_noisy.cpu().numpy()[:, 0, 0], got_clean.cpu().numpy()[:, 0, 0]
    for n in range(len(rows)):
        wn, wc = oa.vad_pair(clean[items[n]["src"]], items[n], noise_pool, irs, length)
        np.testing.assert_array_equal(gc[n], wc, err_msg=f"clean of item {n}")
        # the target is the clean clip with the INPUT's gain, shifted, cut and clipped (data.py:541-562)
        it = items[n]
        direct = np.clip(oa.shift(clean[it["src"]] * np.float32(it["gain"]), it["shift"]), -1, 1)[it["cut"]: it["cut"] + length]
        np.testing.assert_array_equal(gc[n], direct.astype(np.float32), err_msg=f"target of item {n}")
        err = np.abs(gn[n].astype(np.float64) - wn).max()
        assert err <= 1e-5, f"noisy of item {n}: max |diff| {err:.3e}"
        assert np.abs(gn[n]).max() <= 1.0
    # a second call with fewer items reuses nothing from the first
    sub = t.assign_rows(items[[1, 4, 5]].copy())
    n2, c2 = t.apply(sub)
    np.testing.assert_array_equal(n2.cpu().numpy()[:, 0, 0], gn[[1, 4, 5]])
    np.testing.assert_array_equal(c2.cpu().numpy()[:, 0, 0], gc[[1, 4, 5]])


@pytest.mark.gpu
def test_drawn_batch_matches_oracle_and_is_reproducible():
    aug = _aug()
    pools, noisy, clean, noises, irs = _pools(aug, N=12, seed=11)
    t = aug.VADTrainAugment(pools, 8000, p=0.5, seed=5)
    idx = list(range(12))
    items = t.draw(idx, epoch=2)
    a_n, a_c = t.apply(items)
    b_n, b_c = t(idx, epoch=2)
    assert torch.equal(a_n, b_n) and torch.equal(a_c, b_c)
    noise_pool = np.concatenate(noises)
    for n in range(12):
        wn, wc = oa.vad_pair(clean[n], items[n], noise_pool, irs, 8000)
        assert np.abs(a_n[n, 0, 0].cpu().numpy().astype(np.float64) - wn).max() <= 1e-5
        np.testing.assert_array_equal(a_c[n, 0, 0].cpu().numpy(), wc)
    # the base NoiseDataset chain (data.py:410-427): separate noisy clips, target = Cut(Shift(clean)) without gain
    pools2, noisy2, clean2, noises2, irs2 = _pools(aug, N=12, seed=11, paired=True)
    p_n, p_c = aug.VADTrainAugment(pools2, 8000, p=0.5, seed=5).apply(items)
    for n in range(12):
        wn, wc = oa.noise_pair(noisy2[n], clean2[n], items[n], noise_pool, irs2, 8000)
        assert np.abs(p_n[n, 0, 0].cpu().numpy().astype(np.float64) - wn).max() <= 1e-5
        np.testing.assert_array_equal(p_c[n, 0, 0].cpu().numpy(), wc)
    # validation transforms cut the middle (data.py:39-41)
    v = aug.VADTrainAugment(pools, 8000, p=0.5, is_val=True, seed=5).draw(idx)
    assert (v["cut"] == (pools.clip_length - 8000) // 2).all()


@pytest.mark.gpu
def test_full_size_properties():
    """64 clips of 10 s with a 1 s impulse response: linearity and the unit impulse (no oracle at this size)."""
    aug = _aug()
    g = torch.Generator(device="cuda").manual_seed(0)
    L = 160000
    x = torch.randn(64, L, generator=g, device="cuda") * 0.1
    rng = np.random.RandomState(1)
    h = (rng.randn(16000) * np.exp(-np.arange(16000) / 3000.0)).astype(np.float32)
    unit = np.zeros(1500, dtype=np.float32)
    unit[1499] = 1.0
    pool = aug.ImpulseResponsePool([h, unit])
    idx = torch.zeros(64, dtype=torch.int32, device="cuda")
    y = aug.fir_convolve(x, idx, pool)
    y2 = aug.fir_convolve(x * 0.5 + torch.roll(x, 1, 0) * 0.25, idx, pool)
    lin = y * 0.5 + torch.roll(y, 1, 0) * 0.25
    assert (y2 - lin).abs().max().item() <= 3e-5 * y.abs().max().item()
    yu = aug.fir_convolve(x, idx + 1, pool)
    assert (yu[:, 1499:] - x[:, :L - 1499]).abs().max().item() <= 3e-6
    assert yu[:, :1499].abs().max().item() <= 3e-6
    # one row against float64 direct convolution
    want = np.convolve(x[5].cpu().numpy().astype(np.float64), h.astype(np.float64))[:L]
    _close(y[5].cpu().numpy(), want)


def test_draw_is_layout_independent_and_in_range():
    aug = _aug()
    L = 160000
    pools = SimpleNamespace(clip_length=L, noise_lengths=np.array([L + 500, 9000, 3 * L]),
                            noise_offsets=np.array([0, L + 500, L + 9500]), irs=SimpleNamespace(count=4))
    t = aug.VADTrainAugment(pools, 8000, p=0.5, seed=1)
    whole = t.draw(list(range(40)), epoch=3)
    for i in (0, 7, 39):
        one = t.draw([i], epoch=3)[0]
        for k in ("src", "shift", "gain", "snr_db", "noise_off", "noise_len", "ir", "cut", "invert"):
            assert one[k] == whole[i][k]
    assert (np.abs(whole["shift"]) <= L).all() and (whole["cut"] >= 0).all() and (whole["cut"] + 8000 <= L).all()
    assert (whole["gain"] >= 10 ** (-15 / 20) - 1e-6).all() and (whole["gain"] <= 10 ** (5 / 20) + 1e-6).all()
    assert (whole["snr_db"] >= -5).all() and (whole["snr_db"] <= 20).all()
    assert (whole["noise_len"] >= 1).all() and (whole["noise_len"] <= L).all()
    has = whole["ir"] >= 0
    assert 5 <= has.sum() <= 35
    assert sorted(whole["ir_row"][has]) == list(range(has.sum())) and (whole["ir_row"][~has] == -1).all()
    assert t.draw([3], epoch=4)[0]["cut"] != whole[3]["cut"] or t.draw([3], epoch=4)[0]["gain"] != whole[3]["gain"]
