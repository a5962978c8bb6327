"""File-based front end of the dataset synthesiser: the reference's `build_audio` / `gen_audio` / `main_gen`
(process/process.py:78-306) over 16 kHz PCM wav files, with the arithmetic on the GPU.

`build_audio` strings source files together (random crop of a file that is longer than what is still missing, a gap of
silence between files, clipped files skipped, at most MAXTRIES of them); `gen_audio` repeats it until the activity
detector accepts the result; `main_gen` mixes every (clean, noise) pair at a drawn SNR and writes
`clean_fileid_N.wav`, `noise_fileid_N.wav` and `<clean>_<noise>_snr<S>_tl<T>_fileid_N.wav`.  Everything random is drawn
from the global `numpy.random` / `random` generators with the reference's calls in the reference's order, so the same
seeds and the same files give the same choices.  The activity detector (audiolib.py:251-289) and the mixer
(audiolib.py:176-248) run as HIP kernels (`noise_amd.mixer`); files are read and written by `noise_amd.wavio`.
Resampling (process.py:127-130) is not part of this build: a source file at another rate raises.
"""
import os
import random
import sys
from pathlib import Path

import numpy as np

from . import mixer, wavio

MAXTRIES = 50       # process.py:27
MAXFILELEN = 100    # process.py:28


def is_clipped(audio, clipping_threshold=0.99):
    """audiolib.py:18-19."""
    return bool(np.any(np.abs(audio) > clipping_threshold))


def _read(path, fs):
    try:
        return wavio.read_wav(path, fs).astype(np.float64)
    except ValueError as e:
        if "Hz" in str(e):
            raise NotImplementedError(f"{path}: resampling to {fs} Hz is not part of this build") from e
        return None     # "Audio type not supported" (audiolib.py:46-48): the caller moves on to the next file


def build_audio(is_clean, params, index, audio_samples_length=-1):
    """process.py:78-165: (audio float64, files used, clipped files, index of the last file tried)."""
    fs = params["fs"]
    if audio_samples_length == -1:
        audio_samples_length = int(params["audio_length"] * fs)
    if is_clean:
        source_files, idx = params["cleanfilenames"], index
    elif "noisefilenames" in params:
        source_files, idx = params["noisefilenames"], index
    else:   # noise in per-category directories: a random category, a random starting file (process.py:98-112)
        noisedirs = params["noisedirs"]
        idx_n_dir = np.random.randint(0, np.size(noisedirs))
        source_files = [str(f) for f in Path(noisedirs[idx_n_dir]).rglob(params["audioformat"])]
        random.shuffle(source_files)
        idx = np.random.randint(0, np.size(source_files))
    silence = np.zeros(int(fs * params["silence_length"]))
    pieces, files_used, clipped_files = [], [], []
    remaining, tries_left = audio_samples_length, MAXTRIES
    while remaining > 0 and tries_left > 0:
        idx = (idx + 1) % np.size(source_files)
        audio = _read(source_files[idx], fs)
        if audio is None:
            sys.stderr.write("WARNING: Cannot read file: %s\n" % source_files[idx])
            continue
        # a file longer than what is missing is cropped at a random position, except clean files of a test set
        if len(audio) > remaining and (not is_clean or not params["is_test_set"]):
            start = np.random.randint(0, len(audio) - remaining)
            audio = audio[start:start + remaining]
        if is_clipped(audio):
            clipped_files.append(source_files[idx])
            tries_left -= 1
            continue
        files_used.append(source_files[idx])
        pieces.append(audio)
        remaining -= len(audio)
        if remaining > 0:
            gap = min(remaining, len(silence))
            pieces.append(silence[:gap])
            remaining -= gap
    if tries_left == 0 and not is_clean and "noisedirs" in params:
        print("There are not enough non-clipped files in the " + noisedirs[idx_n_dir] +
              " directory to complete the audio build")
        return [], [], clipped_files, idx
    return (np.concatenate(pieces) if pieces else np.zeros(0)), files_used, clipped_files, idx


def gen_audio(is_clean, params, index, audio_samples_length=-1, device="cuda"):
    """process.py:168-199: build_audio until the result is long enough and active enough."""
    clipped_files, low_activity_files = [], []
    if audio_samples_length == -1:
        audio_samples_length = int(params["audio_length"] * params["fs"])
    threshold = params["clean_activity_threshold"] if is_clean else params["noise_activity_threshold"]
    while True:
        audio, source_files, new_clipped, index = build_audio(is_clean, params, index, audio_samples_length)
        clipped_files += new_clipped
        if len(audio) < audio_samples_length:
            continue
        if threshold == 0.0:
            break
        if mixer.activitydetector(audio, device=device) > threshold:
            break
        low_activity_files += source_files
    return audio, source_files, clipped_files, low_activity_files, index


def _joined(files):
    return "-".join(f[:-4].split(os.path.sep)[-1] for f in files)[:MAXFILELEN]


def main_gen(params, device="cuda"):
    """process.py:202-306: generates files fileindex_start..fileindex_end; returns the six bookkeeping lists."""
    clean_sf_all, clean_cf_all, clean_laf_all, noise_sf_all, noise_cf_all, noise_laf_all = [], [], [], [], [], []
    clean_index = noise_index = 0
    file_num = params["fileindex_start"]
    while file_num <= params["fileindex_end"]:
        clean, clean_sf, clean_cf, clean_laf, clean_index = gen_audio(True, params, clean_index, device=device)
        noise, noise_sf, noise_cf, noise_laf, noise_index = gen_audio(False, params, noise_index, len(clean), device=device)
        clean_cf_all += clean_cf
        clean_laf_all += clean_laf
        noise_cf_all += noise_cf
        noise_laf_all += noise_laf
        snr = np.random.randint(params["snr_lower"], params["snr_upper"]) if params["randomize_snr"] else params["snr"]
        clean_snr, noise_snr, noisy_snr, target_level = mixer.segmental_snr_mixer(params, clean, noise, snr, device=device)
        if is_clipped(clean_snr) or is_clipped(noise_snr) or is_clipped(noisy_snr):
            print("Warning: File #" + str(file_num) + " has unexpected clipping, returning without writing audio to disk")
            continue
        clean_sf_all += clean_sf
        noise_sf_all += noise_sf
        noisy_name = (_joined(clean_sf) + "_" + _joined(noise_sf) + "_snr" + str(snr) + "_tl" + str(target_level) +
                      "_fileid_" + str(file_num) + ".wav")
        for directory, name, audio in ((params["noisyspeech_dir"], noisy_name, noisy_snr),
                                       (params["clean_proc_dir"], "clean_fileid_" + str(file_num) + ".wav", clean_snr),
                                       (params["noise_proc_dir"], "noise_fileid_" + str(file_num) + ".wav", noise_snr)):
            try:
                wavio.write_wav(os.path.join(directory, name), audio, params["fs"])
            except Exception as e:   # the reference reports and goes on (process.py:290-294)
                print(str(e))
        file_num += 1
    return clean_sf_all, clean_cf_all, clean_laf_all, noise_sf_all, noise_cf_all, noise_laf_all
