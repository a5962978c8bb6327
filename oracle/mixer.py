"""CPU oracle for the per-utterance SNR mixer (TEST INFRASTRUCTURE ONLY).

Vectorised numpy float64 restatement of the reference's dataset-synthesis maths.
Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Parity pin: golden vectors produced by running the reference's own functions in
the build container (tests/golden/make_golden.py).

Reference lines followed (process/audiolib.py):
  segmental_snr_mixer   :176-216   (the random level draw is an argument here)
  active_rms            :219-248   (threshold -50 on a linear RMS is always passed,
                                    so every 100 ms window is "active")
  normalize_segmental_rms :30-35, is_clipped :18-19
  activitydetector      :251-289
"""
import numpy as np

EPS = np.finfo(float).eps


def active_rms(clean, noise, fs=16000, energy_thresh=-50):
    win = int(fs * 100 / 1000)
    n = len(noise)
    keep = np.zeros(n, dtype=bool)
    for s in range(0, n, win):
        e = min(s + win, n)
        if np.sqrt(np.mean(noise[s:e] ** 2)) > energy_thresh:
            keep[s:e] = True
    if keep.any():
        return np.sqrt(np.mean(clean[keep] ** 2)), np.sqrt(np.mean(noise[keep] ** 2))
    return EPS, EPS


def segmental_snr_mixer(clean, noise, snr, level, target_level=-25, clipping_threshold=0.99):
    """Returns (clean, noise, noisy, level).  `level` is the integer dBFS the reference draws
    with np.random.randint(target_level_lower, target_level_upper)."""
    if len(clean) > len(noise):
        noise = np.append(noise, np.zeros(len(clean) - len(noise)))
    else:
        clean = np.append(clean, np.zeros(len(noise) - len(clean)))
    clean = clean / (np.max(np.abs(clean)) + EPS)
    noise = noise / (np.max(np.abs(noise)) + EPS)
    rmsclean, rmsnoise = active_rms(clean, noise)
    clean = clean * (10 ** (target_level / 20) / (rmsclean + EPS))
    noise = noise * (10 ** (target_level / 20) / (rmsnoise + EPS))
    noisescalar = rmsclean / (10 ** (snr / 20)) / (rmsnoise + EPS)
    noise = noise * noisescalar
    noisy = clean + noise
    rmsnoisy = np.sqrt(np.mean(noisy ** 2))
    scalarnoisy = 10 ** (level / 20) / (rmsnoisy + EPS)
    noisy = noisy * scalarnoisy
    clean = clean * scalarnoisy
    noise = noise * scalarnoisy
    if np.any(np.abs(noisy) > clipping_threshold):
        peak = np.max(np.abs(noisy)) / (clipping_threshold - EPS)
        noisy = noisy / peak
        clean = clean / peak
        noise = noise / peak
        level = int(20 * np.log10(scalarnoisy / peak * (rmsnoisy + EPS)))
    return clean, noise, noisy, level


def activitydetector(audio, fs=16000, energy_thresh=0.13, target_level=-25):
    rms = np.sqrt(np.mean(audio ** 2))
    audio = audio * (10 ** (target_level / 20) / (rms + EPS))
    win = int(fs * 50 / 1000)
    a, b, alpha_rel, alpha_att = -1.0, 0.2, 0.05, 0.8
    prev = 0.0
    active = 0
    cnt = 0
    for s in range(0, len(audio), win):
        w = audio[s:s + win]
        frame_rms = 20 * np.log10(np.sum(w ** 2) + EPS)
        prob = 1.0 / (1 + np.exp(-(a + b * frame_rms)))
        if prob > prev:
            sm = prob * alpha_att + prev * (1 - alpha_att)
        else:
            sm = prob * alpha_rel + prev * (1 - alpha_rel)
        if sm > energy_thresh:
            active += 1
        prev = prob            # the reference carries the un-smoothed probability
        cnt += 1
    return active / cnt
