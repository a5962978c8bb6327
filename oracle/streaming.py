"""TEST INFRASTRUCTURE ONLY (imported by tests/): CPU restatement of the reference's streaming protocol.

The browser demo (demo/src/worker.js:5-66) feeds the exported model (train/export.py:15-23: fixed input
(1,1,1,160*50)) from a microphone stream:

  * `maxSize = 50` chunks of `frames` samples form the model window, `size = 20` chunks are new per model run
    (worker.js:5-6, 16-17);
  * every incoming chunk is written at chunk slot `maxSize - size + offset` (worker.js:20-25);
  * after `size` chunks the whole window runs through the model, the LAST `size` chunks of the result are queued
    for output, and the window is shifted left by `size` chunks (worker.js:28-49);
  * every call returns the oldest queued output chunk, or nothing while the queue is empty (worker.js:52-54).

`forward` maps a float32 window (max_size*frames,) to an output of the same length.
Parity: the reference holds no fixture for this protocol (it only exists as the JavaScript above) - pinned by the
restatement's own properties (delay, hop) and by the model forward, which the golden vectors pin.
"""
import numpy as np


class WindowStream:
    def __init__(self, forward, frames=160, max_size=50, size=20):
        self.forward, self.frames, self.max_size, self.size = forward, frames, max_size, size
        self.buffer = np.zeros(frames * max_size, dtype=np.float32)   # worker.js:17
        self.offset = 0
        self.outputs = []

    def push(self, chunk):
        f, keep = self.frames, self.max_size - self.size
        chunk = np.asarray(chunk, dtype=np.float32)
        assert chunk.shape == (f,)
        self.buffer[(keep + self.offset) * f:(keep + self.offset + 1) * f] = chunk      # worker.js:20-25
        self.offset += 1
        if self.offset == self.size:                                                    # worker.js:28
            result = np.asarray(self.forward(self.buffer.copy()), dtype=np.float32)
            assert result.shape == self.buffer.shape
            out = result[keep * f:]                                                     # worker.js:42
            self.outputs.extend(out[i:i + f] for i in range(0, out.size, f))            # worker.js:43-45
            self.buffer[:keep * f] = self.buffer[self.size * f:].copy()                 # worker.js:46-48
            self.offset = 0
        return self.outputs.pop(0) if self.outputs else None                            # worker.js:52-54
