"""Sliding-window streaming inference for many concurrent streams on one GPU.

Device-resident counterpart of the reference's browser worker (demo/src/worker.js:5-66) around the exported model
(train/export.py:15-23, input (1,1,1,160*50)): a window of `max_size` chunks per stream, `size` new chunks per
model run, the last `size` chunks of every result are emitted, the window then moves on by `size` chunks.  The
worker handles one microphone; here N streams share every launch (window buffers (N, max_size*frames) in HBM, one
batched forward per hop, replayed from a HIP graph so a hop costs one graph launch instead of ~60 kernel launches).

    s = StreamingDenoiser(model.eval(), n_streams=1024)
    out = s.push_block(block)        # block (N, size*frames) of new samples -> (N, size*frames) enhanced samples
    out = s.push(chunk)              # chunk (N, frames): the worker's chunk-by-chunk protocol; None while warming up

Results are views of a buffer that the next model run overwrites.  Streams are independent: sharding streams over
GPUs needs no collective (noise_amd.dist.shard_range).
"""
import torch

from . import _lib


class StreamingDenoiser:
    def __init__(self, model, n_streams, frames=160, max_size=50, size=20, graph=True):
        if model.training:
            raise RuntimeError("StreamingDenoiser needs an eval-mode model: call .eval() first")
        if not (0 < size <= max_size and frames > 0 and n_streams > 0):
            raise ValueError("need 0 < size <= max_size, frames > 0, n_streams > 0")
        self.model, self.N = model, n_streams
        self.frames, self.max_size, self.size = frames, max_size, size
        dev = model.stft.forward_basis.device
        if dev.type != "cuda":
            raise RuntimeError("StreamingDenoiser runs on the GPU only: move the model to a CUDA device first")
        L = frames * max_size
        hop = model.stft.hop_length
        if hop * (L // hop) != L:
            raise ValueError(f"a window of {L} samples does not come back whole from the iSTFT at hop {hop} "
                             f"({hop * (L // hop)} samples): choose frames*max_size as a multiple of the hop")
        # two window buffers: the left shift after a run is a copy from one to the other (the ranges overlap)
        self._buf = torch.zeros(n_streams, L, device=dev, dtype=torch.float32)
        self._alt = torch.zeros_like(self._buf)
        self._offset = 0
        self._pending = 0          # output chunks of the last run not yet handed out
        self._out = None           # (N, size*frames) view of the last result
        self._run_model = None
        self._graph = graph

    def reset(self):
        self._buf.zero_()
        self._offset = 0
        self._pending = 0
        self._out = None

    # ---------------------------------------------------------------------------------------------
    def _forward(self):
        window = self._buf.view(self.N, 1, 1, -1)
        if self._run_model is None:
            if self._graph:
                self._run_model = self.model.graphed_forward(window)
            else:
                def eager(x):
                    with torch.no_grad():
                        return self.model(x)
                self._run_model = eager
        out = self._run_model(window)
        keep = (self.max_size - self.size) * self.frames
        self._out = out.view(self.N, -1)[:, keep:]                        # worker.js:42
        self._alt[:, :keep].copy_(self._buf[:, self.size * self.frames:])   # worker.js:46-48
        self._buf, self._alt = self._alt, self._buf
        self._offset = 0
        self._pending = self.size

    def push(self, chunk):
        """One chunk (N, frames) per stream in, one enhanced chunk out (None during the first size-1 calls)."""
        _lib.require_cuda(chunk)
        if tuple(chunk.shape) != (self.N, self.frames):
            raise ValueError(f"chunk must be ({self.N}, {self.frames}), got {tuple(chunk.shape)}")
        at = (self.max_size - self.size + self._offset) * self.frames     # worker.js:20-25
        self._buf[:, at:at + self.frames].copy_(chunk)
        self._offset += 1
        if self._offset == self.size:
            self._forward()
        if self._pending == 0:
            return None
        k = self.size - self._pending
        self._pending -= 1
        return self._out[:, k * self.frames:(k + 1) * self.frames]

    def push_block(self, block):
        """size*frames new samples per stream in, the same number of enhanced samples out: one model run."""
        _lib.require_cuda(block)
        n = self.size * self.frames
        if tuple(block.shape) != (self.N, n):
            raise ValueError(f"block must be ({self.N}, {n}), got {tuple(block.shape)}")
        if self._offset != 0 or self._pending != 0:
            raise RuntimeError("push_block needs a stream at a hop boundary (no partial hop pushed, no chunk pending)")
        self._buf[:, (self.max_size - self.size) * self.frames:].copy_(block)
        self._forward()
        self._pending = 0
        return self._out
