"""Synthetic batch source for the solver.  The reference's datasets / PIL augmentation (datasets/, utils/transforms.py)
are upstream of the hot path and out of scope; what the path needs from them is the batch layout
(solvers/vince_solver.py:191-199,215-223): ``data`` and ``queue_data`` float32 NCHW already normalised, ``batch_type``,
``batch_size``, ``data_source``, ``num_frames``.  Frames are generated on the device."""
import torch


class SyntheticFrames:
    def __init__(self, batch_size, height, width, num_frames=1, device="cuda", seed=1000, data_source="SYN",
                 iterations=None, rank=0, world=1):
        self.batch_size, self.h, self.w, self.num_frames = batch_size, height, width, num_frames
        self.device, self.seed, self.data_source = device, seed, data_source
        self.iterations = iterations
        self.rank, self.world = rank, world
        self.count = 0
        self._gen = torch.Generator(device=device) if str(device) != "cpu" else torch.Generator()

    def __call__(self, loader_id=0):
        if self.iterations is not None and self.count >= self.iterations:
            self.count = 0
            return None   # end of a loader epoch -> the solver refills the queue (vince_solver.py:379-384)
        self._gen.manual_seed(self.seed + self.count * self.world + self.rank)
        shape = (self.batch_size, 3, self.h, self.w)
        data = torch.randn(shape, generator=self._gen, device=self.device)
        queue_data = data + 0.25 * torch.randn(shape, generator=self._gen, device=self.device)
        self.count += 1
        return {"data": data, "queue_data": queue_data, "batch_type": "images", "batch_size": self.batch_size,
                "data_source": self.data_source, "num_frames": self.num_frames}


class AugmentedFrames:
    """Raw uint8 frames + the GPU input stage: what the reference's dataset + ``--transform`` pair produces
    (r2v2_dataset.py:55-99 applies the transform to every frame of a sample; the solver reads ``data`` / ``queue_data``,
    vince_solver.py:191-199), with the pixel work done by csrc/augment.hip on the training GPU instead of 40 PIL workers.

    ``pool``: uint8 [P, Hs, Ws, 3] frames on the device (a synthetic pool here; a decoder / loader would refill it);
    ``transform``: an instance from ``vince_amd.utils.transforms``.  Each call takes the next ``batch_size`` frames of the pool
    and returns two independently augmented views -- query ``data`` and key ``queue_data`` -- as ``U8Frames`` handles."""

    def __init__(self, pool, transform, batch_size, num_frames=1, data_source="SYN", iterations=None, side_stream=False):
        if pool.dtype != torch.uint8 or pool.dim() != 4 or pool.shape[-1] != 3:
            raise ValueError("AugmentedFrames: pool must be uint8 [P, Hs, Ws, 3]")
        self.pool, self.transform, self.batch_size = pool.contiguous(), transform, batch_size
        self.num_frames, self.data_source, self.iterations = num_frames, data_source, iterations
        self.count = 0
        # side_stream: run the augmentation kernels on their own stream.  Off by default -- measured at BASELINE config 3 the
        # three launches cost their 1.0 ms of device time when they sit on the consumer's stream (29.0 -> 30.0 ms/step) but
        # 2.9 ms from a fifth concurrent stream (the step already keeps four busy; VINCE_AUG_STREAM=1 forces it on)
        import os
        side_stream = side_stream or os.environ.get("VINCE_AUG_STREAM", "0") == "1"
        self._stream = torch.cuda.Stream(device=pool.device) if (side_stream and pool.is_cuda) else None

    def __call__(self, loader_id=0):
        if self.iterations is not None and self.count >= self.iterations:
            self.count = 0
            return None
        b, p = self.batch_size, self.pool.shape[0]
        idx = (torch.arange(b) + self.count * b) % p
        self.count += 1
        import numpy as np
        src = np.concatenate([idx.numpy(), idx.numpy()]).astype(np.int64)
        params = self.transform.draw(2 * b, tuple(self.pool.shape[1:3]), src_index=src)
        if self._stream is None:
            views = self.transform.apply(self.pool, params)
        else:
            consumer = torch.cuda.current_stream(self.pool.device)
            with torch.cuda.stream(self._stream):     # (a caller that refills `pool` must order that against this stream)
                views = self.transform.apply(self.pool, params)
            consumer.wait_stream(self._stream)          # device-side join, no host wait
            for t in (views.frames, views.flip) + (views.blur or ()):
                if t is not None:
                    t.record_stream(consumer)
        return {"data": views[0:b], "queue_data": views[b:2 * b], "batch_type": "images", "batch_size": b,
                "data_source": self.data_source, "num_frames": self.num_frames}
