"""File-based front end of the synthesiser (noise_amd/synth_files.py) against what the reference's own `build_audio` /
`gen_audio` (process/process.py:78-199) produced for the same wav files and seeds (tests/golden/g7_synth_files.npz,
written by tests/golden/make_golden.py)."""
import os
import random
import wave

import numpy as np
import pytest

G7 = os.path.join(os.path.dirname(__file__), "golden", "g7_synth_files.npz")


def _materialise(g7, root):
    params = dict(fs=16000, silence_length=0.05, audio_length=1.5, is_test_set=False, audioformat="*.wav",
                  clean_activity_threshold=0.0, noise_activity_threshold=0.0)
    for kind in ("clean", "noise"):
        paths = []
        for name in g7[f"{kind}_names"]:
            p = os.path.join(root, str(name))
            with wave.open(p, "wb") as f:
                f.setnchannels(1)
                f.setsampwidth(2)
                f.setframerate(16000)
                f.writeframes(g7[f"src_{name}"].astype("<i2").tobytes())
            paths.append(p)
        params[f"{kind}filenames"] = paths
    return params


def _base(files):
    return [os.path.basename(f) for f in files]


def test_build_audio_matches_reference(tmp_path):
    from noise_amd import synth_files
    g7 = np.load(G7)
    params = _materialise(g7, str(tmp_path))
    case = 0
    while f"build{case}_args" in g7.files:
        is_clean, index, length, test_set, seed = (int(v) for v in g7[f"build{case}_args"])
        params["is_test_set"] = bool(test_set)
        np.random.seed(seed)
        random.seed(seed)
        audio, used, clipped, idx = synth_files.build_audio(bool(is_clean), params, index, length)
        np.testing.assert_array_equal(audio, g7[f"build{case}_audio"])
        assert _base(used) == list(g7[f"build{case}_used"]) and _base(clipped) == list(g7[f"build{case}_clipped"])
        assert idx == int(g7[f"build{case}_idx"])
        case += 1
    assert case == 5


def test_other_sampling_rate_is_refused(tmp_path):
    from noise_amd import synth_files, wavio
    p = str(tmp_path / "x.wav")
    wavio.write_wav(p, np.zeros(100), 8000)
    params = dict(fs=16000, silence_length=0.0, audio_length=0.01, is_test_set=False, cleanfilenames=[p])
    with pytest.raises(NotImplementedError):
        synth_files.build_audio(True, params, 0)


@pytest.mark.gpu
def test_gen_audio_and_main_gen(tmp_path):
    from noise_amd import synth_files, wavio
    g7 = np.load(G7)
    params = _materialise(g7, str(tmp_path))
    params["clean_activity_threshold"] = 0.6
    np.random.seed(77)
    random.seed(77)
    audio, used, clipped, low, idx = synth_files.gen_audio(True, params, 2, 16000)
    np.testing.assert_array_equal(audio, g7["gen_audio"])
    assert _base(used) == list(g7["gen_used"]) and _base(low) == list(g7["gen_low_activity"])
    assert _base(clipped) == list(g7["gen_clipped"]) and idx == int(g7["gen_idx"])
    from noise_amd import mixer
    assert abs(mixer.activitydetector(audio) - float(g7["gen_activity"])) <= 1e-12
    # the generation loop: three files, written in the reference's layout and readable by the data module's pairing
    out = tmp_path / "out"
    params.update(fileindex_start=0, fileindex_end=2, randomize_snr=True, snr_lower=-5, snr_upper=20, snr=5,
                  target_level_lower=-35, target_level_upper=-15, noisyspeech_dir=str(out / "noisy"),
                  clean_proc_dir=str(out / "clean"), noise_proc_dir=str(out / "noise"), noise_activity_threshold=0.0,
                  clean_activity_threshold=0.3)
    np.random.seed(5)
    random.seed(5)
    lists = synth_files.main_gen(params)
    assert len(lists) == 6 and len(lists[0]) >= 3
    pairs = wavio.pair_files(str(out))
    assert sorted(os.path.basename(c) for _, c in pairs) == [f"clean_fileid_{i}.wav" for i in range(3)]
    for noisy_path, clean_path in pairs:
        name = os.path.basename(noisy_path)
        assert "_snr" in name and "_tl-" in name and name.endswith(".wav")
        y, c = wavio.read_wav(noisy_path), wavio.read_wav(clean_path)
        n = wavio.read_wav(os.path.join(str(out), "noise", os.path.basename(clean_path).replace("clean_", "noise_")))
        assert len(y) == len(c) == len(n) == 24000 and np.abs(y).max() <= 0.99
        assert np.abs(y - (c + n)).max() <= 2.5 / 32768      # three independently rounded 16-bit files
