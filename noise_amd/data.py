"""The reference's `VADNoiseDataModule` (train/data.py:224-233, 430-437, 541-562) with the work of its dataloader
workers moved to the GPU.

Same constructor and batch contract: a batch is `(noisy, clean, names)` with `noisy`, `clean` float32
`(B,1,1,length)` clipped to [-1,1] and `names` a list of strings.  Two sources:

* `data_dir` with the reference's layout (`train|val/{noisy,clean,noise,ir}/*.wav`, 16 kHz 16-bit PCM): the clips,
  background noises and impulse responses are read once into HBM (`augment.AugmentPools`) and every batch goes through
  the augmentation chain of train/data.py:439-484 on the device (`augment.VADTrainAugment`); pairs are dealt to ranks
  with DistributedSampler semantics (shuffled per epoch with a shared seed, index i of the order goes to rank
  i % world_size).
* no `data_dir` (or one that does not exist): synthetic audio generated on the device (16 kHz, speech-like 4 Hz
  envelope, white noise at a random SNR in [-5, 20) dB).
"""
import math
import os

import numpy as np
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

        self._aug = {}
        self.epoch = 0

    def setup(self, stage=None):
        """data.py:339-366: builds the train and validation sets when `data_dir` holds them."""
        if not self.data_dir or not os.path.isdir(str(self.data_dir)):
            return
        from . import augment, wavio
        for split in ("train", "val"):
            root = os.path.join(str(self.data_dir), split)
            pairs = wavio.pair_files(root)
            if not pairs:
                continue
            if split == "train" and self.num_samples > 0:
                pairs = pairs[: self.num_samples]                       # data.py:399-400
            noisy = [wavio.read_wav(a, self.sampling_rate) for a, _ in pairs]
            clean = [wavio.read_wav(b, self.sampling_rate) for _, b in pairs]
            L = min(min(len(x) for x in noisy), min(len(x) for x in clean))
            noises = [wavio.read_wav(f, self.sampling_rate) for f in wavio.list_wavs(os.path.join(root, "noise"))]
            irs = [wavio.read_wav(f, self.sampling_rate) for f in wavio.list_wavs(os.path.join(root, "ir"))]
            pools = augment.AugmentPools(np.stack([x[:L] for x in noisy]), np.stack([x[:L] for x in clean]), noises, irs,
                                         self.device)
            names = [os.path.splitext(os.path.basename(b))[0][6:] for _, b in pairs]      # data.py:427
            self._aug[split] = (augment.VADTrainAugment(pools, self.length, self.p, is_val=(split == "val")), names)

    def _augmented_loader(self, split, shuffle):
        aug, names = self._aug[split]
        n = len(names)
        order = np.random.default_rng([20240, self.epoch]).permutation(n) if shuffle else np.arange(n)
        mine = order[self.rank::self.world_size]
        for i in range(0, len(mine), self.batch_size):
            idx = mine[i:i + self.batch_size]
            noisy, clean = aug(idx.tolist(), epoch=self.epoch)
            yield noisy, clean, [names[j] for j in idx]
        if shuffle:
            self.epoch += 1

    def _loader(self, base_seed):
        n_batches = max(1, self.num_samples // (self.batch_size * self.world_size))
        for i in range(n_batches):
            # DistributedSampler semantics: every rank draws a different slice of the epoch
            yield synth_batch(self.batch_size, self.length, self.device, base_seed + i * self.world_size + self.rank,
                              self.sampling_rate)

    def train_dataloader(self):
        return self._augmented_loader("train", True) if "train" in self._aug else self._loader(1234)

    def val_dataloader(self):
        return self._augmented_loader("val", False) if "val" in self._aug else self._loader(987654)
