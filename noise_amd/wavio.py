"""16 kHz PCM wav files of the dataset layout the reference reads and writes (host side, standard library only).

The synthesiser writes `clean_fileid_N.wav`, `noise_fileid_N.wav` and `<sources>_snr<S>_tl<T>_fileid_N.wav`
(process/process.py:268-285, through soundfile's default 16-bit PCM); the training data module pairs every file under
`noisy/` with `clean/clean_fileid_N.wav` by the number after the last underscore (train/data.py:305-315) and loads
both as float32 mono (librosa.load, :412-413).  `soundfile` and `librosa` are not needed for that format: 16-bit PCM
is int16 / 32768 on the way in and round(x * 32768) clipped to int16 on the way out (libsndfile's PCM_16 scaling).
"""
import os
import wave

import numpy as np


def read_wav(path, sampling_rate=16000):
    """float32 mono samples in [-1, 1) of a 16-bit PCM wav; channels are averaged (audiolib.py:55-57)."""
    with wave.open(str(path), "rb") as f:
        if f.getsampwidth() != 2:
            raise ValueError(f"{path}: only 16-bit PCM is served (sample width {f.getsampwidth()} bytes)")
        if f.getframerate() != sampling_rate:
            raise ValueError(f"{path}: {f.getframerate()} Hz, expected {sampling_rate} (resampling is not part of this build)")
        ch = f.getnchannels()
        raw = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2")
    x = raw.astype(np.float32) / 32768.0
    return x.reshape(-1, ch).mean(axis=1).astype(np.float32) if ch > 1 else x


def write_wav(path, audio, sampling_rate=16000):
    """audio (float, [-1, 1]) -> 16-bit PCM mono wav; the directory is created like audiowrite does (audiolib.py:89-95)."""
    path = os.path.abspath(str(path))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    pcm = np.clip(np.rint(np.asarray(audio, dtype=np.float64) * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sampling_rate)
        f.writeframes(pcm.tobytes())


def pair_files(data_dir):
    """[(noisy_path, clean_path)] of `data_dir`/noisy and `data_dir`/clean, paired by file id (data.py:305-315).
    Unlike the reference nothing is renamed on disk; noisy files without a clean partner are skipped."""
    pairs = []
    noisy_dir, clean_dir = os.path.join(data_dir, "noisy"), os.path.join(data_dir, "clean")
    if not os.path.isdir(noisy_dir):
        return pairs
    for root, _, files in sorted(os.walk(noisy_dir)):
        for name in sorted(files):
            if not name.endswith(".wav"):
                continue
            fileid = "fileid_" + os.path.splitext(name)[0].split("_")[-1]
            clean = os.path.join(clean_dir, f"clean_{fileid}.wav")
            if os.path.exists(clean):
                pairs.append((os.path.join(root, name), clean))
    return pairs


def list_wavs(directory):
    out = []
    for root, _, files in sorted(os.walk(directory)):
        out += [os.path.join(root, n) for n in sorted(files) if n.endswith(".wav")]
    return out
