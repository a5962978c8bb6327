"""The CPU restatement of the waveform augmentation against the fixture the reference's own functions wrote
(tests/golden/g6_augment.npz: ImpulseResponse.apply, JustNoise.apply, Cut.apply of train/data.py and add_pyreverb of
process/process.py, run on synthetic sounds by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import augment as oa

G6 = os.path.join(os.path.dirname(__file__), "golden", "g6_augment.npz")


@pytest.fixture(scope="module")
def g6():
    return np.load(G6)


@pytest.mark.parametrize("name", ["ir_short", "ir_room", "ir_hall"])
def test_impulse_response_and_reverb(g6, name):
    speech, ir = g6["speech"], g6[name]
    np.testing.assert_array_equal(oa.impulse_response(speech, ir), g6[f"impulse_response_{name}"])
    np.testing.assert_array_equal(oa.add_pyreverb(speech, ir), g6[f"add_pyreverb_{name}"])


@pytest.mark.parametrize("tag,sound", [("long", "noise_long"), ("short", "noise_short")])
def test_noise_scaling_and_tiling(g6, tag, sound):
    speech = g6["speech"]
    start, snr = g6[f"noise_{tag}_params"]
    start = int(start)
    seg = g6[sound][start:start + len(speech)]
    got = oa.scaled_tiled_noise(speech, seg, snr)
    np.testing.assert_allclose(got, g6[f"noise_{tag}_scaled"], rtol=1e-6, atol=0)
    assert len(got) == len(speech)


def test_cut(g6):
    speech = g6["speech"]
    np.testing.assert_array_equal(oa.cut(speech, int(g6["cut_train_offset"]), 8000), g6["cut_train"])
    assert int(g6["cut_val_offset"]) == (len(speech) - 8000) // 2
    np.testing.assert_array_equal(oa.cut(speech, int(g6["cut_val_offset"]), 8000), g6["cut_val"])


def test_shift_gain_semantics():
    x = np.arange(1, 9, dtype=np.float32)
    np.testing.assert_array_equal(oa.shift(x, 3), [0, 0, 0, 1, 2, 3, 4, 5])
    np.testing.assert_array_equal(oa.shift(x, -2), [3, 4, 5, 6, 7, 8, 0, 0])
    np.testing.assert_array_equal(oa.shift(x, 0), x)
    assert oa.gain(x, 0.5).dtype == np.float32
