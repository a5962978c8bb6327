"""Waveform augmentation of the VAD training data module on the GPU (train/data.py:439-461, 541-562).

The reference builds every training pair on dataloader workers from ONE clean clip (`VADNoiseDataset.__getitem__`,
train/data.py:541-562): the network input is the clean clip through Gain -> Shift -> AddBackgroundNoise ->
ImpulseResponse (p = 0.5) -> Cut -> PolarityInversion -> Clip, then the same parameters are frozen and the target is
the clean clip through the transforms named Gain, Shift, Cut and Clip only, i.e. Clip(Cut(Shift(Gain(clean)))) with
the gain of the input.  (The base `NoiseDataset`, train/data.py:410-427, reads separate noisy / clean files and keeps
only Shift and Cut for the target: `AugmentPools(noisy=...)`.)  Here the source clips, the background noises and the impulse responses live in HBM
(`AugmentPools`), the parameters of a batch are drawn on the host (`VADTrainAugment.draw`, distributions of
train/data.py:441-458) and the arithmetic runs in the kernels of csrc/augment.hip through the C-ABI; the output is
the reference's batch tuple `(noisy, clean)` of shape (B, 1, 1, length) on the device.

`add_pyreverb` and `impulse_response` are drop-ins (numpy in, numpy out) for process/process.py:69-75 and
`ImpulseResponse.apply` (train/data.py:186-209).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from ._lib import _I, _L, _P, ptr

_lib.register({
    "noise_fir_spectra": [_P, _L, _I, _L, _I, _P, _P, _P, _P],
    "noise_fir_apply": [_P, _I, _L, _P, _P, _P, _P, _I, _P, _P, _P, _L, _P, _P],
    "noise_aug_mix": [_P, _I, _P, _P, _L, _L, _P, _L, _P, _P, _P, _P, _I, _I, _P],
    "noise_aug_finish": [_P, _P, _I, _P, _L, _P, _P, _P, _I, _P],
})
for _name in ("noise_fir_block", "noise_fir_blocks", "noise_fir_partials", "noise_aug_chunks"):
    getattr(_lib.lib, _name).restype = ctypes.c_int
_lib.lib.noise_fir_block.argtypes = []
for _name in ("noise_fir_blocks", "noise_fir_partials", "noise_aug_chunks"):
    getattr(_lib.lib, _name).argtypes = [_L]

# struct noise_aug_item of include/noise_hip.h
ITEM_DTYPE = np.dtype([("src", np.int32), ("shift", np.int32), ("gain", np.float32), ("snr_db", np.float32),
                       ("noise_off", np.int64), ("noise_len", np.int32), ("ir", np.int32), ("ir_row", np.int32),
                       ("cut", np.int32), ("invert", np.int32)], align=True)
assert ITEM_DTYPE.itemsize == 48

_TW = {}


def _twiddles(device):
    """e^{-2 pi i j/1024}, j < 1024, and e^{-2 pi i k/2048}, k <= 1024, of the 2048-point packed real transform."""
    key = str(device)
    if key not in _TW:
        a1 = -2.0 * math.pi * torch.arange(1024, dtype=torch.float64) / 1024
        a2 = -2.0 * math.pi * torch.arange(1025, dtype=torch.float64) / 2048
        _TW[key] = tuple(torch.stack([torch.cos(a), torch.sin(a)], dim=1).float().contiguous().to(device) for a in (a1, a2))
    return _TW[key]


def fir_blocks(L):
    return _lib.lib.noise_fir_blocks(int(L))


def block_spectra(x, L=None, filter=False):
    """x (R, >= L) float32 CUDA rows -> spectra (R, fir_blocks(L), 1024, 2) of their 2048-sample segments at hop 1024."""
    _lib.require_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    L = int(x.shape[1] if L is None else L)
    R = x.shape[0]
    tw1, tw2 = _twiddles(x.device)
    spec = torch.empty(R, fir_blocks(L), 1024, 2, device=x.device, dtype=torch.float32)
    _lib.call("noise_fir_spectra", ptr(x), x.stride(0), R, L, int(bool(filter)), ptr(tw1), ptr(tw2), ptr(spec), _lib.stream(),
              nbytes=4 * R * L + spec.numel() * 4)
    return spec


class ImpulseResponsePool:
    """Impulse responses kept as partition spectra: one 2048-point transform per 1024 taps."""

    def __init__(self, impulse_responses, device="cuda"):
        if len(impulse_responses) == 0:
            raise ValueError("at least one impulse response is needed")
        device = torch.device(device)
        specs, first, nblk = [], [], []
        total = 0
        self.lengths = []
        for ir in impulse_responses:
            h = torch.as_tensor(np.asarray(ir, dtype=np.float32)).reshape(1, -1).to(device)
            if h.shape[1] == 0:
                raise ValueError("empty impulse response")
            s = block_spectra(h, filter=True)[0]
            specs.append(s)
            first.append(total)
            nblk.append(s.shape[0])
            total += s.shape[0]
            self.lengths.append(h.shape[1])
        self.spec = torch.cat(specs, 0).contiguous()
        self.first_block = torch.tensor(first, dtype=torch.int64, device=device)
        self.nblk = torch.tensor(nblk, dtype=torch.int32, device=device)
        self.count = len(nblk)
        self.device = device


def fir_convolve(x, ir_index, pool, out=None, partial=None, L=None):
    """y[r] = fftconvolve(x[r], ir[ir_index[r]], mode="full")[:L] for the rows of x (R, L) float32 CUDA."""
    _lib.require_cuda(x, ir_index)
    L = int(x.shape[1] if L is None else L)
    R = x.shape[0]
    spec = block_spectra(x, L)
    y = torch.empty(R, L, device=x.device, dtype=torch.float32) if out is None else out
    tw1, tw2 = _twiddles(x.device)
    ir_index = ir_index.to(torch.int32).contiguous()
    nblk = fir_blocks(L)
    _lib.call("noise_fir_apply", ptr(spec), R, L, ptr(ir_index), ptr(pool.spec), ptr(pool.first_block), ptr(pool.nblk),
              pool.count, ptr(tw1), ptr(tw2), ptr(y), y.stride(0), ptr(partial), _lib.stream(),
              nbytes=8 * R * nblk * 1024 + 4 * R * L)
    return y


def add_pyreverb(clean_speech, rir, device="cuda"):
    """Drop-in for process/process.py:69-75: full convolution truncated to the length of the speech (float32)."""
    x = torch.as_tensor(np.asarray(clean_speech, dtype=np.float32)).reshape(1, -1).to(device)
    pool = ImpulseResponsePool([rir], device)
    y = fir_convolve(x, torch.zeros(1, dtype=torch.int32, device=x.device), pool)
    return y[0].cpu().numpy()


def impulse_response(samples, ir, device="cuda"):
    """Drop-in for ImpulseResponse.apply (train/data.py:186-209) with the impulse response given as an array:
    convolve, truncate, restore the RMS of the input, clip to [-1, 1]."""
    x = torch.as_tensor(np.asarray(samples, dtype=np.float32)).reshape(1, -1).to(device)
    L = x.shape[1]
    pool = ImpulseResponsePool([ir], device)
    partial = torch.empty(_lib.lib.noise_fir_partials(L), device=x.device, dtype=torch.float64)
    y = fir_convolve(x, torch.zeros(1, dtype=torch.int32, device=x.device), pool, partial=partial)
    in_rms = torch.sqrt((x.double() ** 2).mean()).float()
    out_rms = torch.sqrt(partial.sum() / L).float()
    scale = torch.where(out_rms > 0, in_rms / out_rms, torch.ones_like(in_rms))
    return (y[0] * scale).clamp_(-1.0, 1.0).cpu().numpy()


class AugmentPools:
    """Device-resident sources: clean clips of one length, background noises, impulse responses.

    `noisy` (optional, same shape as `clean`): separate already-noisy clips for the input chain, as the base
    `NoiseDataset` reads them (train/data.py:410-427); without it the input chain starts from the clean clip
    (`VADNoiseDataset`, train/data.py:541-562) and the target carries the input's gain and the final Clip."""

    def __init__(self, clean, noise_sounds, impulse_responses, device="cuda", noisy=None):
        device = torch.device(device)
        self.clean = torch.as_tensor(clean, dtype=torch.float32).to(device).contiguous()
        self.paired = noisy is not None
        self.noisy = torch.as_tensor(noisy, dtype=torch.float32).to(device).contiguous() if self.paired else self.clean
        if self.clean.dim() != 2 or self.noisy.shape != self.clean.shape:
            raise ValueError("clean (and noisy) must be (N, L) arrays of one shape")
        if len(noise_sounds) == 0:
            raise ValueError("at least one background noise is needed (data.py:75)")
        self.noise_lengths = np.array([len(s) for s in noise_sounds], dtype=np.int64)
        self.noise_offsets = np.concatenate([[0], np.cumsum(self.noise_lengths)[:-1]]).astype(np.int64)
        self.noise = torch.as_tensor(np.concatenate([np.asarray(s, dtype=np.float32) for s in noise_sounds])).to(device)
        self.irs = ImpulseResponsePool(impulse_responses, device)
        self.device = device

    @property
    def clip_length(self):
        return self.clean.shape[1]


class VADTrainAugment:
    """`VADNoiseDataModule.train_transforms` / `.val_transforms` (train/data.py:439-484) for whole batches."""

    def __init__(self, pools, length, p=0.5, is_val=False, seed=0):
        if length <= 0 or length > pools.clip_length:
            raise ValueError(f"length {length} outside 1..{pools.clip_length}")
        self.pools, self.length, self.p, self.is_val, self.seed = pools, int(length), float(p), bool(is_val), int(seed)

    def draw(self, indices, epoch=0):
        """Parameters of the pairs `indices`, one generator per (seed, epoch, index): the result for a pair does not
        depend on the batch it is drawn in or on the number of ranks."""
        pools, L = self.pools, self.pools.clip_length
        items = np.zeros(len(indices), dtype=ITEM_DTYPE)
        for n, idx in enumerate(indices):
            g = np.random.default_rng([self.seed, int(epoch), int(idx)])
            it = items[n]
            it["src"] = idx
            it["gain"] = np.float32(10.0 ** (g.uniform(-15.0, 5.0) / 20.0))            # Gain(-15, 5, p=1), data.py:441
            if g.random() < self.p:                                                      # Shift(-1, 1, rollover=False), :442-448
                it["shift"] = int(round(g.uniform(-1.0, 1.0) * L))
            it["snr_db"] = g.uniform(-5.0, 20.0)                                         # AddBackgroundNoise(-5, 20, p=1), :449-454
            sound = int(g.integers(len(pools.noise_lengths)))
            n_noise = int(pools.noise_lengths[sound])
            start = int(g.integers(0, max(0, n_noise - L - 1) + 1))                      # data.py:109-117
            it["noise_off"] = pools.noise_offsets[sound] + start
            it["noise_len"] = min(L, n_noise - start)
            it["ir"] = int(g.integers(pools.irs.count)) if g.random() < 0.5 else -1      # ImpulseResponse(p=0.5), :455
            it["cut"] = max((L - self.length) // 2, 0) if self.is_val else int(g.integers(0, max(L - self.length, 1)))  # :39-46
            it["invert"] = int(g.random() < self.p)                                      # PolarityInversion(p), :457
        return self.assign_rows(items)

    @staticmethod
    def assign_rows(items):
        """Numbers the items that take an impulse response: their row of the convolution workspaces."""
        has = items["ir"] >= 0
        items["ir_row"] = -1
        items["ir_row"][has] = np.arange(int(has.sum()), dtype=np.int32)
        return items

    def validate(self, items):
        pools, L = self.pools, self.pools.clip_length
        if items.dtype != ITEM_DTYPE:
            raise TypeError("items must use augment.ITEM_DTYPE")
        ok = ((items["src"] >= 0) & (items["src"] < pools.noisy.shape[0]) & (items["cut"] >= 0)
              & (items["cut"] + self.length <= L) & (items["noise_len"] >= 0) & (items["noise_len"] <= L)
              & (items["noise_off"] >= 0) & (items["noise_off"] + items["noise_len"] <= pools.noise.numel())
              & (items["ir"] < pools.irs.count) & (np.abs(items["shift"].astype(np.int64)) <= L))
        if not ok.all():
            raise ValueError(f"augmentation item {int(np.argmin(ok))} is out of range: {items[int(np.argmin(ok))]}")
        rows = items["ir_row"][items["ir"] >= 0]
        if len(rows) and not np.array_equal(np.sort(rows), np.arange(len(rows))):
            raise ValueError("ir_row must number the items with an impulse response 0..n-1 (assign_rows)")

    def apply(self, items):
        """items (ITEM_DTYPE array) -> (noisy, clean) float32 CUDA tensors of shape (B, 1, 1, length)."""
        self.validate(items)
        pools, L, length = self.pools, self.pools.clip_length, self.length
        dev = pools.device
        B = len(items)
        d_items = torch.from_numpy(items.view(np.uint8).copy()).to(dev)
        with_ir = np.nonzero(items["ir"] >= 0)[0]
        with_ir = with_ir[np.argsort(items["ir_row"][with_ir])]
        R = len(with_ir)
        nch = _lib.lib.noise_aug_chunks(L)
        partial = torch.empty(B * nch * 3, device=dev, dtype=torch.float64)
        mixed = torch.empty(max(R, 1), L, device=dev, dtype=torch.float32)
        noisy = torch.empty(B, 1, 1, length, device=dev, dtype=torch.float32)
        clean = torch.empty(B, 1, 1, length, device=dev, dtype=torch.float32)
        st = _lib.stream()
        _lib.call("noise_aug_mix", ptr(d_items), B, ptr(pools.noisy), ptr(pools.clean), pools.noisy.stride(0), L,
                  ptr(pools.noise), pools.noise.numel(), ptr(partial), ptr(mixed), ptr(noisy), ptr(clean), length,
                  int(not pools.paired), st,
                  nbytes=4 * (2 * B * L + R * L + 2 * B * length))
        if R:
            ir_of_row = torch.from_numpy(items["ir"][with_ir].astype(np.int32)).to(dev)
            item_of_row = torch.from_numpy(with_ir.astype(np.int32)).to(dev)
            rev_partial = torch.empty(R * _lib.lib.noise_fir_partials(L), device=dev, dtype=torch.float64)
            reverb = fir_convolve(mixed, ir_of_row, pools.irs, partial=rev_partial)
            _lib.call("noise_aug_finish", ptr(d_items), ptr(item_of_row), R, ptr(reverb), L, ptr(partial), ptr(rev_partial),
                      ptr(noisy), length, st, nbytes=8 * R * length)
        return noisy, clean

    def __call__(self, indices, epoch=0):
        return self.apply(self.draw(indices, epoch))
