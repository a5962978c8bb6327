"""Streaming window protocol (demo/src/worker.js:5-66): the CPU restatement's own properties, and the GPU driver
against it with the CPU oracle's model forward."""
import os

import numpy as np
import pytest
import torch

from oracle.streaming import WindowStream


def test_oracle_protocol_delay_and_hop():
    """With an identity model the worker returns nothing for size-1 calls, then every chunk size-1 calls late, and
    the model sees windows that advance by `size` chunks."""
    seen = []

    def identity(window):
        seen.append(window.copy())
        return window

    s = WindowStream(identity, frames=4, max_size=5, size=2)
    chunks = [np.full(4, float(i + 1), dtype=np.float32) for i in range(9)]
    outs = [s.push(c) for c in chunks]
    assert outs[0] is None
    for i in range(1, 9):
        np.testing.assert_array_equal(outs[i], chunks[i - 1])
    assert len(seen) == 4
    np.testing.assert_array_equal(seen[0].reshape(5, 4)[:, 0], [0, 0, 0, 1, 2])
    np.testing.assert_array_equal(seen[1].reshape(5, 4)[:, 0], [0, 1, 2, 3, 4])
    np.testing.assert_array_equal(seen[3].reshape(5, 4)[:, 0], [4, 5, 6, 7, 8])


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [True, False], ids=["hipgraph", "eager"])
def test_streaming_matches_oracle(golden_dir, graph):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import spectral as osp
    from noise_amd import StreamingDenoiser
    import test_gpu_model as tgm
    dev = torch.device("cuda:0")
    m = tgm.build(320, 160, dev, golden_dir).eval()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    bufs = osp.stft_buffers(1200, 320, 160)

    def cpu_forward(window):
        with torch.no_grad():
            out = osp.model_forward(torch.from_numpy(window).view(1, 1, 1, -1), sd, bufs, 320, 160)
        out = out[0] if isinstance(out, (tuple, list)) else out
        return out.reshape(-1).numpy()

    N, frames, n_chunks = 3, 160, 65
    gen = torch.Generator().manual_seed(11)
    audio = (torch.randn(N, n_chunks * frames, generator=gen) * 0.1).clamp(-1, 1)
    ref = [WindowStream(cpu_forward) for _ in range(N)]
    s = StreamingDenoiser(m, N, graph=graph)
    produced = 0
    for c in range(n_chunks):
        chunk = audio[:, c * frames:(c + 1) * frames]
        got = s.push(chunk.to(dev))
        want = [r.push(chunk[i].numpy()) for i, r in enumerate(ref)]
        if want[0] is None:
            assert got is None
            continue
        produced += 1
        # fp32 end to end, [-1,1] waveforms: 1e-4 absolute like the eval-forward parity test
        np.testing.assert_allclose(got.cpu().numpy(), np.stack(want), atol=1e-4)
    assert produced == n_chunks - 19

    # block interface: same results as 20 single chunks, at a hop boundary
    s.reset()
    a = StreamingDenoiser(m, N, graph=graph)
    for hop in range(2):
        block = audio[:, hop * 3200:(hop + 1) * 3200].to(dev)
        out_block = a.push_block(block).clone()
        outs = [s.push(block[:, k * frames:(k + 1) * frames]) for k in range(20)]
        first = outs[-1]            # the call that triggers the run returns the first chunk of the result
        assert all(o is None for o in outs[:-1]) or hop > 0
        np.testing.assert_array_equal(first.cpu().numpy(), out_block[:, :frames].cpu().numpy())
    with pytest.raises(ValueError):
        a.push_block(torch.zeros(N, 100, device=dev))
    with pytest.raises(RuntimeError):
        StreamingDenoiser(m.train(), N)
