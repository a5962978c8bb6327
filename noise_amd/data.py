"""Synthetic stand-in for the reference's `VADNoiseDataModule` (train/data.py:224-233, 430-437, 541-562).

Same constructor and batch contract: a batch is `(noisy, clean, names)` with `noisy`, `clean` float32
`(B,1,1,length)` clipped to [-1,1] and `names` a list of strings.  The audio is generated on the device
(16 kHz, speech-like 4 Hz envelope, white noise at a random SNR in [-5, 20) dB) instead of being read
from disk and augmented on CPU workers.
"""
import math

import torch


def synth_batch(batch_size, length, device, seed=1234, sampling_rate=16000):
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.arange(length, dtype=torch.float32) / sampling_rate
    phase = torch.rand(batch_size, 1, generator=g) * 2 * math.pi
    env = 0.5 * (1 - torch.cos(2 * math.pi * 4.0 * t[None, :] + phase))
    clean = (0.05 * torch.randn(batch_size, length, generator=g) * env).clamp(-1, 1)
    silent = torch.rand(batch_size, generator=g) < 0.15          # some all-silent clips -> VAD label 0
    clean[silent] = 0.0
    noise = 0.02 * torch.randn(batch_size, length, generator=g)
    snr = torch.randint(-5, 20, (batch_size, 1), generator=g).float()
    rms_c = clean.pow(2).mean(1, keepdim=True).sqrt().clamp_min(1e-4)
    rms_n = noise.pow(2).mean(1, keepdim=True).sqrt()
    noisy = (clean + noise * rms_c / rms_n * 10 ** (-snr / 20)).clamp(-1, 1)
    names = [f"synthetic_{seed}_{i}" for i in range(batch_size)]
    return (noisy.view(batch_size, 1, 1, length).to(device), clean.view(batch_size, 1, 1, length).to(device), names)


class VADNoiseDataModule:
    def __init__(self, data_dir=None, sampling_rate=16000, length=8000, num_samples=10000, num_workers=0,
                 batch_size=128, p=0.3, device="cuda", rank=0, world_size=1):
        self.data_dir, self.sampling_rate, self.length = data_dir, sampling_rate, length
        self.num_samples, self.num_workers, self.batch_size, self.p = num_samples, num_workers, batch_size, p
        self.device, self.rank, self.world_size = device, rank, world_size

    def setup(self, stage=None):
        pass

    def _loader(self, base_seed):
        n_batches = max(1, self.num_samples // (self.batch_size * self.world_size))
        for i in range(n_batches):
            # DistributedSampler semantics: every rank draws a different slice of the epoch
            yield synth_batch(self.batch_size, self.length, self.device, base_seed + i * self.world_size + self.rank,
                              self.sampling_rate)

    def train_dataloader(self):
        return self._loader(1234)

    def val_dataloader(self):
        return self._loader(987654)
