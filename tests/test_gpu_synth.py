"""Sharded synthesis driver (process/process.py:200-285 + STFT features): the result for a file index does not
depend on the number of ranks, the mix equals the CPU oracle's, the features equal the STFT of the mix."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_generate_shard_layout_independent_and_matches_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from noise_amd import STFT, synth
    from oracle import mixer as omix
    dev = torch.device("cuda:0")
    st = STFT(200, 320, 160, 320, "hann").to(dev)
    got = {}

    def sink(b, first, out):
        got[b] = {k: (v.cpu() if torch.is_tensor(v) else [(f[0].cpu(), f[1].cpu()) for f in v]) for k, v in out.items()}

    n_files, L, block = 70, 16000, 16          # 5 blocks, the last one ragged (6 files)
    one = synth.generate_shard(n_files, 0, 1, length=L, seed=3, block=block, stfts=[st], sink=sink)
    whole = dict(got)
    assert one["files"] == n_files and one["blocks"] == [0, 1, 2, 3, 4] and one["kept"] + one["clipped"] == n_files
    got.clear()
    parts = [synth.generate_shard(n_files, r, 3, length=L, seed=3, block=block, stfts=[st], sink=sink) for r in range(3)]
    assert sorted(b for p in parts for b in p["blocks"]) == [0, 1, 2, 3, 4]
    assert parts[0]["blocks"] == [0, 3] and parts[1]["blocks"] == [1, 4] and parts[2]["blocks"] == [2]
    for b in whole:      # bit-identical whatever the number of ranks
        assert torch.equal(whole[b]["noisy"], got[b]["noisy"]) and torch.equal(whole[b]["level"], got[b]["level"])
        assert one["checksums"][b] == [p for p in parts if b in p["checksums"]][0]["checksums"][b]
    # the mix against the CPU oracle (float64; only reduction order differs), the features against the STFT of the mix
    b = 4
    clean, noise, _ = synth.clean_noise_block(b, 6, L, 3, dev)
    for i in range(6):
        c, n, y, lv = omix.segmental_snr_mixer(clean[i].cpu().numpy(), noise[i].cpu().numpy(),
                                               int(whole[b]["snr"][i]), int(_level_draw(b, 6, 3, dev)[i]))
        np.testing.assert_allclose(whole[b]["noisy"][i].numpy(), y, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(whole[b]["clean"][i].numpy(), c, rtol=1e-12, atol=1e-15)
        assert int(whole[b]["level"][i]) == lv
        assert bool(whole[b]["kept"][i]) == (max(np.abs(c).max(), np.abs(n).max(), np.abs(y).max()) <= 0.99)
    lm, ph = st.forward(whole[b]["noisy"].to(torch.float32).view(6, 1, 1, L).to(dev))
    assert torch.equal(lm.cpu(), whole[b]["features"][0][0]) and torch.equal(ph.cpu(), whole[b]["features"][0][1])


def _level_draw(b, n, seed, dev):
    """The level draw of block b, replayed from the block's generator (after the audio and the SNRs)."""
    from noise_amd import synth
    _, _, g = synth.clean_noise_block(b, n, 16000, seed, dev)
    torch.randint(-5, 20, (n,), generator=g, device=dev, dtype=torch.int32)
    return torch.randint(-35, -15, (n,), generator=g, device=dev, dtype=torch.int32).cpu()
