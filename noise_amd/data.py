"""The reference's `VADNoiseDataModule` (train/data.py:224-233, 430-437, 486-562) with the work of its dataloader
workers moved to the GPU.

Same constructor and batch contract: a batch is `(noisy, clean, names)` with `noisy`, `clean` float32
`(B,1,1,length)` clipped to [-1,1] and `names` a list of strings.  Two sources:

* `data_dir` with the reference's layout (`train|val/{clean,noise,ir}/*.wav`, 16 kHz 16-bit PCM;
  `VADNoiseDataModule.get_files`, train/data.py:486-488, reads `clean/` only): the clips, background noises and impulse
  responses are read once into HBM (`augment.AugmentPools`) and every batch goes through the augmentation chain of
  train/data.py:439-484 on the device (`augment.VADTrainAugment`): input = the whole chain on the clean clip, target =
  the clean clip with the same gain, shift and cut (train/data.py:541-562).  Pairs are dealt to ranks with
  DistributedSampler semantics: the order is shuffled per epoch with a shared seed, padded with its own head to a
  multiple of world_size (every rank yields the same number of batches, so no rank leaves the gradient all-reduce
  early), and index i of it goes to rank i % world_size.
* `data_dir=None`: synthetic audio generated on the device (16 kHz, speech-like 4 Hz envelope, white noise at a random
  SNR in [-5, 20) dB).  A `data_dir` that is given but holds no clips raises instead of training on synthetic audio.
"""
import math
import os
import warnings

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
                 batch_size=128, p=0.3, device="cuda", rank=0, world_size=1, seed=0):
        self.data_dir, self.sampling_rate, self.length = data_dir, sampling_rate, length
        self.num_samples, self.num_workers, self.batch_size, self.p = num_samples, num_workers, batch_size, p
        self.device, self.rank, self.world_size, self.seed = device, rank, world_size, seed

        self._aug = {}
        self.epoch = 0

    def set_epoch(self, epoch):
        """Epoch of the next `train_dataloader()`: selects the shuffle and the augmentation draws (all ranks must
        agree on it, as with DistributedSampler.set_epoch)."""
        self.epoch = int(epoch)

    def setup(self, stage=None):
        """train/data.py:490-517: builds the train and validation sets from `data_dir`/{train,val}/clean."""
        if self.data_dir is None:
            return
        if not os.path.isdir(str(self.data_dir)):
            raise FileNotFoundError(f"data_dir {self.data_dir!r} does not exist (pass data_dir=None for synthetic audio)")
        from . import augment, wavio
        for split in ("train", "val"):
            root = os.path.join(str(self.data_dir), split)
            files = wavio.list_wavs(os.path.join(root, "clean"))             # data.py:486-488
            if not files:
                if split == "train":
                    raise FileNotFoundError(f"no wav files under {root}/clean: the VAD data module reads "
                                            "data_dir/{train,val}/{clean,noise,ir}")
                continue
            if split == "train":
                # data.py:529-531: the file list is shuffled before it is cut to num_samples (here with a seeded
                # generator shared by all ranks instead of the process-global `random` state)
                order = np.random.default_rng([self.seed, 7]).permutation(len(files))
                files = [files[i] for i in order]
                if self.num_samples > 0:
                    files = files[: self.num_samples]
            clips, kept = [], []
            for f in files:
                x = wavio.read_wav(f, self.sampling_rate)
                if len(x) < self.length:
                    warnings.warn(f"{f}: {len(x)} samples, shorter than length={self.length}; skipped")
                    continue
                clips.append(x)
                kept.append(f)
            if not clips:
                raise ValueError(f"every clip under {root}/clean is shorter than length={self.length}")
            L = min(len(x) for x in clips)
            if any(len(x) != L for x in clips):
                # the device pools hold clips of one length: longer clips keep their first L samples
                warnings.warn(f"{root}/clean: clips of {min(map(len, clips))}..{max(map(len, clips))} samples are "
                              f"cut to the shortest ({L})")
            noises = [wavio.read_wav(f, self.sampling_rate) for f in wavio.list_wavs(os.path.join(root, "noise"))]
            irs = [wavio.read_wav(f, self.sampling_rate) for f in wavio.list_wavs(os.path.join(root, "ir"))]
            pools = augment.AugmentPools(np.stack([x[:L] for x in clips]), noises, irs, self.device)
            names = [os.path.splitext(os.path.basename(f))[0][6:] for f in kept]          # data.py:562
            self._aug[split] = (augment.VADTrainAugment(pools, self.length, self.p, is_val=(split == "val"),
                                                        seed=self.seed), names)

    def rank_indices(self, n, shuffle, epoch):
        """DistributedSampler: shuffled order, padded with its head to a multiple of world_size, strided by rank."""
        order = np.random.default_rng([20240, self.seed, int(epoch)]).permutation(n) if shuffle else np.arange(n)
        total = -(-n // self.world_size) * self.world_size
        if total > n:
            order = np.concatenate([order, np.resize(order, total - n)])
        return order[self.rank::self.world_size]

    def _augmented_loader(self, split, shuffle, epoch):
        aug, names = self._aug[split]
        mine = self.rank_indices(len(names), shuffle, epoch)
        for i in range(0, len(mine), self.batch_size):
            idx = mine[i:i + self.batch_size]
            noisy, clean = aug(idx.tolist(), epoch=epoch)
            yield noisy, clean, [names[j] for j in idx]

    def _loader(self, base_seed):
        n_batches = max(1, self.num_samples // (self.batch_size * self.world_size))
        for i in range(n_batches):
            # DistributedSampler semantics: every rank draws a different slice of the epoch
            yield synth_batch(self.batch_size, self.length, self.device, base_seed + i * self.world_size + self.rank,
                              self.sampling_rate)

    def train_dataloader(self):
        """One epoch.  The epoch counter advances when the loader is created (not when it is exhausted), so a loop
        that stops early still sees a new shuffle next time; `set_epoch` overrides it."""
        if "train" not in self._aug:
            return self._loader(1234)
        epoch, self.epoch = self.epoch, self.epoch + 1
        return self._augmented_loader("train", True, epoch)

    def val_dataloader(self):
        return self._augmented_loader("val", False, 0) if "val" in self._aug else self._loader(987654)
