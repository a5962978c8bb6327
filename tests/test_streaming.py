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
@pytest.mark.parametrize("graph,N", [(True, 3), (False, 3), (True, 1)], ids=["hipgraph", "eager", "one-stream"])
def test_streaming_matches_oracle(golden_dir, graph, N):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import spectral as osp
    from noise_amd import StreamingDenoiser
    import test_gpu_model as tgm
    dev = torch.device("cuda:0")
    m = tgm.build(320, 160, dev, golden_dir).eval()
    with torch.no_grad():
        m.model.vad.bias.fill_(3.0)       # keep the VAD gate (models.py:754) open: a closed gate silences the window
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    bufs = osp.stft_buffers(1200, 320, 160)

    def cpu_forward(window):
        with torch.no_grad():
            out = osp.model_forward(torch.from_numpy(window).view(1, 1, 1, -1), sd, bufs, 320, 160)
        out = out[0] if isinstance(out, (tuple, list)) else out
        return out.reshape(-1).numpy()

    frames, n_chunks = 160, 65
    gen = torch.Generator().manual_seed(11)
    audio = (torch.randn(N, n_chunks * frames, generator=gen) * 0.1).clamp(-1, 1)
    ref = [WindowStream(cpu_forward) for _ in range(N)]
    s = StreamingDenoiser(m, N, graph=graph)
    produced, nonzero = 0, False
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
        nonzero = nonzero or float(np.abs(np.stack(want)).max()) > 1e-3
    assert produced == n_chunks - 19 and nonzero

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


@pytest.mark.gpu
def test_streaming_bf16_path(golden_dir):
    """The same protocol on the block-fused bf16 model path (HIP-graph replay of 13 launches per hop): every emitted
    block against the fp32 oracle at the bf16 path's stated waveform tolerance (SNR >= 33 dB with the fixture's
    untrained weights), and bit-identical to the model's own bf16 forward of the same window."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import spectral as osp
    from noise_amd import StreamingDenoiser
    import test_gpu_model as tgm
    dev = torch.device("cuda:0")
    m = tgm.build(320, 160, dev, golden_dir).eval()
    with torch.no_grad():
        m.model.vad.bias.fill_(3.0)       # gate open for most windows
    m.inference_dtype = "bf16"
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    bufs = osp.stft_buffers(1200, 320, 160)
    N = 4
    gen = torch.Generator().manual_seed(12)
    env = 0.5 - 0.5 * torch.cos(2 * np.pi * 4.0 * torch.arange(5 * 3200) / 16000.0)
    audio = (0.05 * torch.randn(N, 5 * 3200, generator=gen) * env + 0.02 * torch.randn(N, 5 * 3200, generator=gen)).clamp(-1, 1)
    s = StreamingDenoiser(m, N)
    window = torch.zeros(N, 8000)
    refs, gots = [], []
    for hop in range(5):
        block = audio[:, hop * 3200:(hop + 1) * 3200]
        window = torch.cat([window[:, 3200:], block], dim=1)
        got = s.push_block(block.to(dev)).cpu()
        with torch.no_grad():
            ref = osp.model_forward(window.view(N, 1, 1, -1), sd, bufs, 320, 160).reshape(N, -1)[:, -3200:]
            own = m(window.view(N, 1, 1, -1).to(dev)).reshape(N, -1)[:, -3200:].cpu()
        assert torch.equal(got, own)
        # the VAD gate (models.py:754) silences whole windows: the same windows must be silent on both sides
        closed = ref.abs().amax(dim=1) == 0
        assert torch.equal(closed, got.abs().amax(dim=1) == 0)
        refs.append(ref.double())
        gots.append(got.double())
    ref, got = torch.cat(refs, dim=1), torch.cat(gots, dim=1)
    assert ref.abs().max() > 0, "the fixture must open the gate for some window"
    snr = 10 * torch.log10(ref.pow(2).sum() / (got - ref).pow(2).sum().clamp_min(1e-30))
    assert snr >= 33.0, float(snr)
