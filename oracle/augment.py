"""CPU restatement (numpy float32, scipy fftconvolve) of the VAD data module's waveform augmentation.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and the smoke check); the product path never calls it.

Follows, in the reference: train/data.py:439-461 (the chain and its order), :541-562 (VADNoiseDataset: input and
target both start from the clean clip; the target is transformed with the frozen parameters and only by Gain, Shift,
Cut, Clip), :410-427 (NoiseDataset: separate noisy file, target through Shift and Cut only), :30-57 (Cut), :186-209 (ImpulseResponse.apply),
:119-157 (JustNoise.apply: the reference's own restatement of AddBackgroundNoise's scaling and tiling, which
AddBackgroundNoise finishes with `samples + noise_sound`); process/process.py:69-75 (add_pyreverb).

Gain, Shift, AddBackgroundNoise, PolarityInversion and Clip themselves are `audiomentations` classes (listed in
train/requirements.txt without a version; the library is not installed here).  Their published arithmetic is restated
below: Gain multiplies by 10^(dB/20); Shift(rollover=False, fade_duration=0) is np.roll with the wrapped samples set
to zero; PolarityInversion negates; Clip is np.clip(-1, 1); calculate_rms = sqrt(mean(square(x))),
calculate_desired_noise_rms = clean_rms / 10^(snr/20).  The reference holds no fixture for them: parity of those five
one-liners is unpinned; `cut`, `impulse_response`, `scaled_tiled_noise` and `add_pyreverb` are pinned by
tests/golden/g6_augment.npz, written by the reference's own functions.
"""
import numpy as np
from scipy.signal import fftconvolve


def calculate_rms(samples):
    return np.sqrt(np.mean(np.square(samples), axis=-1))


def calculate_desired_noise_rms(clean_rms, snr):
    return clean_rms / (10 ** (float(snr) / 20))


def gain(samples, amplitude_ratio):
    return samples * np.float32(amplitude_ratio)


def shift(samples, num_places_to_shift):
    n = int(num_places_to_shift)
    out = np.roll(samples, n, axis=-1)
    if n > 0:
        out[..., :n] = 0.0
    elif n < 0:
        out[..., n:] = 0.0
    return out


def scaled_tiled_noise(samples, noise_sound, snr_db):
    """data.py:133-157 for the noise segment already cut out (data.py:126-128); None when the noise is silent."""
    noise_rms = calculate_rms(noise_sound)
    if noise_rms < 1e-9:
        return None
    clean_rms = calculate_rms(samples)
    desired_noise_rms = calculate_desired_noise_rms(clean_rms, snr_db)
    noise_sound = noise_sound * (desired_noise_rms / noise_rms)
    num_samples = len(samples)
    while len(noise_sound) < num_samples:
        noise_sound = np.concatenate((noise_sound, noise_sound))
    return noise_sound[0:num_samples].astype(np.float32)


def add_background_noise(samples, noise_sound, snr_db):
    noise = scaled_tiled_noise(samples, noise_sound, snr_db)
    return samples if noise is None else samples + noise


def add_pyreverb(clean_speech, rir):
    """process.py:69-75."""
    reverb_speech = fftconvolve(clean_speech, rir, mode="full")
    return reverb_speech[0: clean_speech.shape[0]]


def impulse_response(samples, ir):
    """data.py:190-209."""
    reverb_samples = fftconvolve(samples, ir, mode="full")
    reverb_samples = reverb_samples[: len(samples)]
    input_rms = np.sqrt(np.mean(samples ** 2))
    output_rms = np.sqrt(np.mean(reverb_samples ** 2))
    scaling_factor = input_rms / output_rms if output_rms > 0 else 1.0
    reverb_samples = reverb_samples * scaling_factor
    return reverb_samples.clip(-1.0, 1.0)


def cut(samples, offset, length):
    """data.py:48-56."""
    return samples[offset: offset + length] if length > 0 else samples


def _input_chain(x, item, noise_pool, impulse_responses, length):
    x = gain(np.asarray(x, dtype=np.float32), item["gain"])
    x = shift(x, item["shift"])
    if item["noise_len"] > 0:
        seg = noise_pool[int(item["noise_off"]): int(item["noise_off"]) + int(item["noise_len"])]
        x = add_background_noise(x, seg, float(item["snr_db"]))
    if item["ir"] >= 0:
        x = impulse_response(x, impulse_responses[int(item["ir"])])
    x = cut(x, int(item["cut"]), length)
    if item["invert"]:
        x = -x
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def vad_pair(clean, item, noise_pool, impulse_responses, length):
    """One (input, target) pair of VADNoiseDataset (data.py:541-562) with the parameters of `item` (fields of
    noise_aug_item): input = the whole chain on the clean clip, target = Clip(Cut(Shift(Gain(clean))))."""
    clean = np.asarray(clean, dtype=np.float32)
    x = _input_chain(clean, item, noise_pool, impulse_responses, length)
    c = np.clip(cut(shift(gain(clean, item["gain"]), item["shift"]), int(item["cut"]), length), -1.0, 1.0)
    return x, c.astype(np.float32)


def noise_pair(noisy, clean, item, noise_pool, impulse_responses, length):
    """One pair of the base NoiseDataset (data.py:410-427): the input chain runs on the separate noisy clip, the frozen
    pass keeps only Shift and Cut for the clean clip."""
    x = _input_chain(noisy, item, noise_pool, impulse_responses, length)
    c = cut(shift(np.asarray(clean, dtype=np.float32), item["shift"]), int(item["cut"]), length)
    return x, c.astype(np.float32)
