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
