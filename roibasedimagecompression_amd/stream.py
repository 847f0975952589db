"""Stream regime (BASELINE.json configs[4]: a long sequence of independent frames on one GPU).

`FrameEncoder.encode_batch` already runs the k-means++ chains of a batch of frames side by side; what is left
of a batch's wall time is largely host work (the numpy side of levels 2-3) during which the GPU idles, and GPU
work of level 1 during which the host idles.  `StreamEncoder` keeps several batches in flight: every lane is a
host thread with its own HIP stream and its own rhccq context, so the host part of one batch hides behind the
device part of another.  Nothing is shared between lanes except the input tensors, which are read only.

The reference has no counterpart (it encodes one image per notebook run); results are those of
`FrameEncoder.encode` frame by frame (tests/test_gpu_frame.py::test_stream_encoder_equals_frame_by_frame)."""
import queue
import threading

import torch

from .frame import FrameEncoder
from .ops import Rhccq


class StreamEncoder:
    def __init__(self, device=0, batch=8, lanes=2):
        if batch < 1 or lanes < 1:
            raise ValueError("batch and lanes must be >= 1")
        self.device, self.batch, self.lanes = int(device), int(batch), int(lanes)
        self._lane_state = [None] * self.lanes           # (stream, Rhccq, FrameEncoder) per lane, kept across run() calls

    def _lane(self, i):
        """the lane's HIP stream, context and encoder (created on first use, inside the lane's stream)"""
        if self._lane_state[i] is None:
            stream = torch.cuda.Stream(self.device)
            with torch.cuda.stream(stream):
                rh = Rhccq(self.device)                   # binds its context to this lane's stream
            self._lane_state[i] = (stream, rh, FrameEncoder(rh))
        return self._lane_state[i]

    def close(self):
        for st in self._lane_state:
            if st is not None:
                st[1].close()
        self._lane_state = [None] * self.lanes

    def run(self, frames):
        """frames: sequence of (rgb uint8[H,W,3] device tensor, [ClassSpec, ...]).  Returns the per-frame
        outputs of FrameEncoder.encode, in order.  The caller's stream must have finished writing the inputs."""
        frames = list(frames)
        batches = [frames[i:i + self.batch] for i in range(0, len(frames), self.batch)]
        results = [None] * len(batches)
        todo = queue.Queue()
        for j in range(len(batches)):
            todo.put(j)
        errors = []
        torch.cuda.synchronize(self.device)

        def lane(i):
            try:
                torch.cuda.set_device(self.device)
                stream, rh, enc = self._lane(i)
                with torch.cuda.stream(stream):
                    while not errors:
                        try:
                            j = todo.get_nowait()
                        except queue.Empty:
                            break
                        b = batches[j]
                        results[j] = enc.encode_batch(b) if len(b) > 1 else [enc.encode(*b[0])]
                        stream.synchronize()
            except BaseException as e:                        # surfaced to the caller below
                errors.append(e)

        threads = [threading.Thread(target=lane, args=(i,), name=f"rhccq-lane{i}") for i in range(min(self.lanes, max(len(batches), 1)))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return [out for r in results for out in r]
