"""Sharded offline synthesis: noisy/clean mixing plus STFT feature generation, per utterance, no collectives.

GPU counterpart of the reference's generation loop (process/process.py:200-285, `main_gen`): for every file index
draw an SNR (`np.random.randint(snr_lower, snr_upper)`, :231-236), mix with `segmental_snr_mixer` (:238-240), drop
the file when any of the three signals clips (:246-253), keep it otherwise; the training side then turns the audio
into STFT features (train/models.py:460-475).  Here file indices are grouped into blocks of `block` utterances; block
b belongs to rank b % world, and everything random in a block (source audio stand-ins, SNRs, levels) is drawn from a
generator seeded with (seed, b): the result for a given file index does not depend on the number of GPUs.

Source audio is synthetic (the reference reads wav files): `clean_noise_block` stands in for `gen_audio`
(process.py:139-199), which a caller with real data replaces through `source=`.
"""
import torch

from . import mixer
from .stft import STFT


def clean_noise_block(block_index, n, length, seed, device):
    """Synthetic stand-in for gen_audio: speech-like (4 Hz envelope) clean and stationary noise, float64 (n, length)."""
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) << 20) + int(block_index))
    t = torch.arange(length, device=device, dtype=torch.float64) / 16000.0
    phase = torch.rand(n, 1, generator=g, device=device, dtype=torch.float64) * 6.283185307179586
    env = 0.5 - 0.5 * torch.cos(6.283185307179586 * 4.0 * t[None, :] + phase)
    clean = (0.05 * torch.randn(n, length, generator=g, device=device, dtype=torch.float64) * env).clamp_(-1, 1)
    noise = (0.02 * torch.randn(n, length, generator=g, device=device, dtype=torch.float64)).clamp_(-1, 1)
    return clean, noise, g


def generate_shard(n_files, rank=0, world=1, length=160000, seed=0, block=1024, params=None, stfts=(), source=None,
                   sink=None, device="cuda"):
    """Mixes (and featurizes) this rank's share of `n_files` utterances.

    params: the reference's keys snr_lower / snr_upper / target_level_lower / target_level_upper (process.py:231-236,
    audiolib.py:199).  stfts: STFT modules (on `device`) whose features are wanted.  source(block_index, n, length,
    seed, device) -> (clean, noise, generator) replaces the synthetic audio.  sink(block_index, first_file_index, out)
    receives every finished block: out = dict(clean, noise, noisy (n, length) float64, level (n,), snr (n,),
    kept (n,) bool, features [(logmag, phase), ...]).  Returns counters and a per-block checksum of the noisy audio.
    `block` utterances are mixed per launch (one workgroup per utterance: blocks of 1024 x 10 s fill the GPU, blocks of
    256 leave the mixer at 3.6 instead of 4.2 TB/s); everything random in a block is seeded by (seed, block index), so
    results depend on `block` but not on the number of ranks.
    """
    params = params or dict(snr_lower=-5, snr_upper=20, target_level_lower=-35, target_level_upper=-15)
    source = source or clean_noise_block
    device = torch.device(device)
    n_blocks = -(-n_files // block)
    stats = dict(files=0, kept=0, clipped=0, audio_seconds=0.0, blocks=[], checksums={})
    kept_counts, checksums = [], []
    for b in range(rank, n_blocks, world):          # block b belongs to rank b % world: no exchange between ranks
        first = b * block
        n = min(block, n_files - first)
        clean, noise, g = source(b, n, length, seed, device)
        snr = torch.randint(params["snr_lower"], params["snr_upper"], (n,), generator=g, device=device,
                            dtype=torch.int32)
        level = torch.randint(params["target_level_lower"], params["target_level_upper"], (n,), generator=g,
                              device=device, dtype=torch.int32)
        # process.py:246-253 (is_clipped, audiolib.py:18-19): a file with any sample above the threshold is dropped.
        # The peaks of the three signals, the checksum and the float32 mix the STFT reads leave the mixing kernel itself.
        c, nz, y, lv, peaks, y32 = mixer.mix_batch(clean, noise, snr, level, with_peaks=True)
        kept = peaks[:, :3].amax(1) <= 0.99
        feats = [st.forward(y32.view(n, 1, 1, length)) for st in stfts]
        out = dict(clean=c, noise=nz, noisy=y, level=lv, snr=snr, kept=kept, features=feats)
        if sink is not None:
            sink(b, first, out)
        kept_counts.append(kept.sum())
        checksums.append(peaks[:, 3].sum())
        stats["files"] += n
        stats["audio_seconds"] += n * length / 16000.0
        stats["blocks"].append(b)
    if stats["blocks"]:      # one host synchronisation per shard, not one per block
        k = int(torch.stack(kept_counts).sum().item())
        stats["kept"], stats["clipped"] = k, stats["files"] - k
        stats["checksums"] = dict(zip(stats["blocks"], torch.stack(checksums).cpu().tolist()))
    return stats
