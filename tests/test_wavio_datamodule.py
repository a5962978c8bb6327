"""16-bit PCM wav files of the dataset layout (process.py:268-285, data.py:305-315) and the data module that feeds
batches from them through the on-GPU augmentation."""
import os

import numpy as np
import pytest

from noise_amd import wavio


def _write_dataset(root, n=6, L=24000, seed=0):
    rng = np.random.RandomState(seed)
    for split in ("train", "val"):
        for i in range(n):
            clean = (rng.randn(L) * 0.05).astype(np.float32)
            noisy = np.clip(clean + rng.randn(L).astype(np.float32) * 0.02, -1, 1)
            wavio.write_wav(os.path.join(root, split, "clean", f"clean_fileid_{i}.wav"), clean)
            wavio.write_wav(os.path.join(root, split, "noisy", f"book_7_noise_3_snr5_tl-20_fileid_{i}.wav"), noisy)
        wavio.write_wav(os.path.join(root, split, "noise", "fan.wav"), rng.randn(L + 9000) * 0.05)
        wavio.write_wav(os.path.join(root, split, "noise", "click.wav"), rng.randn(2000) * 0.1)
        h = rng.randn(3000) * np.exp(-np.arange(3000) / 500.0) * 0.2
        h[0] = 0.9
        wavio.write_wav(os.path.join(root, split, "ir", "room.wav"), h)


def test_wav_round_trip_and_pairing(tmp_path):
    x = np.array([0.0, 0.5, -0.5, 0.999, -1.0, 1.0, 1e-5, -3.05e-5], dtype=np.float64)
    p = tmp_path / "a" / "b.wav"
    wavio.write_wav(p, x)
    y = wavio.read_wav(p)
    assert y.dtype == np.float32
    np.testing.assert_array_equal(y, np.clip(np.rint(x * 32768), -32768, 32767).astype(np.float32) / 32768)
    assert np.abs(y - np.clip(x, -1, 32767 / 32768)).max() <= 0.5 / 32768 + 1e-9
    with pytest.raises(ValueError):
        wavio.read_wav(p, sampling_rate=8000)
    _write_dataset(str(tmp_path / "ds"), n=3)
    pairs = wavio.pair_files(str(tmp_path / "ds" / "train"))
    assert [os.path.basename(c) for _, c in pairs] == ["clean_fileid_0.wav", "clean_fileid_1.wav", "clean_fileid_2.wav"]
    assert all("_fileid_%d.wav" % i in a for i, (a, _) in enumerate(pairs))
    os.remove(pairs[1][1])
    assert len(wavio.pair_files(str(tmp_path / "ds" / "train"))) == 2


def test_rank_indices_pad_like_distributed_sampler():
    from noise_amd.data import VADNoiseDataModule
    for n, world in ((7, 2), (6, 4), (5, 8), (9, 3)):
        per_rank = [VADNoiseDataModule(None, rank=r, world_size=world).rank_indices(n, True, 3) for r in range(world)]
        assert len({len(p) for p in per_rank}) == 1 and len(per_rank[0]) == -(-n // world)      # equal batch counts
        flat = np.stack(per_rank, 1).reshape(-1)                                                 # the padded order
        assert sorted(set(flat.tolist())) == list(range(n))
        np.testing.assert_array_equal(flat[n:], np.resize(flat[:n], len(flat) - n))              # padded with its own head
    a = VADNoiseDataModule(None, rank=0, world_size=1)
    assert not np.array_equal(a.rank_indices(50, True, 0), a.rank_indices(50, True, 1))
    np.testing.assert_array_equal(a.rank_indices(5, False, 9), np.arange(5))


def test_datamodule_refuses_an_empty_data_dir(tmp_path):
    from noise_amd.data import VADNoiseDataModule
    with pytest.raises(FileNotFoundError):
        VADNoiseDataModule(str(tmp_path / "missing")).setup("fit")
    os.makedirs(tmp_path / "ds" / "train" / "noisy")        # the reference's VAD layout has no noisy/: clean/ is what counts
    with pytest.raises(FileNotFoundError):
        VADNoiseDataModule(str(tmp_path / "ds")).setup("fit")
    assert VADNoiseDataModule(None).setup("fit") is None    # synthetic audio only without a data_dir


@pytest.mark.gpu
def test_datamodule_batches_from_wav_files(tmp_path):
    import torch

    from noise_amd.data import VADNoiseDataModule
    from oracle import augment as oa
    root = str(tmp_path / "ds")
    _write_dataset(root, n=7)
    import shutil
    for split in ("train", "val"):
        shutil.rmtree(os.path.join(root, split, "noisy"))   # VADNoiseDataModule.get_files reads clean/ only (data.py:486-488)
    seen, counts = {}, []
    for rank in (0, 1):
        dm = VADNoiseDataModule(root, length=8000, num_samples=7, batch_size=2, p=0.5, rank=rank, world_size=2)
        dm.setup("fit")
        nb = 0
        for noisy, clean, names in dm.train_dataloader():
            nb += 1
            assert noisy.shape == clean.shape == (len(names), 1, 1, 8000) and noisy.dtype == torch.float32
            assert noisy.abs().max().item() <= 1.0 and clean.abs().max().item() <= 1.0
            for k, name in enumerate(names):
                seen.setdefault(name, []).append((noisy[k, 0, 0].cpu().numpy(), clean[k, 0, 0].cpu().numpy()))
        counts.append(nb)
        assert dm.epoch == 1
        val = list(dm.val_dataloader())
        assert sum(len(b[2]) for b in val) == 4
    assert counts[0] == counts[1] == 2                      # 7 clips on 2 ranks: padded to 8, two batches of 2 each
    assert sorted(seen) == [f"fileid_{i}" for i in range(7)]
    assert sorted(len(v) for v in seen.values()) == [1] * 6 + [2]
    # every pair equals the CPU chain with the parameters its index draws, whatever rank produced it
    dm = VADNoiseDataModule(root, length=8000, num_samples=7, batch_size=7, p=0.5)
    dm.setup("fit")
    aug, names = dm._aug["train"]
    items = aug.draw(list(range(7)), epoch=0)
    pools = aug.pools
    noise_pool = pools.noise.cpu().numpy()
    irs = [wavio.read_wav(os.path.join(root, "train", "ir", "room.wav"))]
    for i, name in enumerate(names):
        wn, wc = oa.vad_pair(pools.clean[i].cpu().numpy(), items[i], noise_pool, irs, 8000)
        for got_n, got_c in seen[name]:
            assert np.abs(got_n.astype(np.float64) - wn).max() <= 1e-5
            np.testing.assert_array_equal(got_c, wc)
    # a clip shorter than `length` is skipped with a warning, not allowed to shorten the whole pool
    wavio.write_wav(os.path.join(root, "train", "clean", "clean_fileid_99.wav"), np.zeros(100))
    dm = VADNoiseDataModule(root, length=8000, num_samples=-1, batch_size=4)
    with pytest.warns(UserWarning, match="shorter than length"):
        dm.setup("fit")
    assert len(dm._aug["train"][1]) == 7 and dm._aug["train"][0].pools.clip_length == 24000
