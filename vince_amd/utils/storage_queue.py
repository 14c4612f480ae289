"""FIFO negative queue (reference utils/storage_queue.py:4-56), same public surface.

``vector_queue`` is a K x D float32 device tensor that is never reallocated by ``enqueue`` (``dequeue`` hands out a
view, storage_queue.py:53).  The ring-buffer index arithmetic (split on wrap, repeated laps for n > K) runs in
``vince_queue_enqueue`` (csrc/misc.hip) and the copies are device-to-device on the current stream.
Keeping one CPU image per slot (39 GB at K=65536, only used for TensorBoard dumps) is optional: ``keep_images``.
"""
import torch

from .. import ops


class StorageQueue:
    def __init__(self, maxsize, feat_size, device=None, dtype=torch.float32, keep_images=False):
        self.maxsize, self.feat_size = maxsize, feat_size
        self.device, self.dtype = device, dtype
        self.keep_images = keep_images
        self.vector_queue = None
        self.clear()

    def __len__(self):
        return self.maxsize

    def clear(self):
        # storage_queue.py:10-12,21-29: rows drawn from a Gaussian and normalised; tail back at slot 0 (K21)
        rows = torch.randn(self.maxsize, self.feat_size, device=self.device, dtype=self.dtype)
        rows = torch.nn.functional.normalize(rows, dim=-1)
        if self.vector_queue is None:
            self.vector_queue = rows
        else:
            self.vector_queue.copy_(rows)    # keep the storage: dequeue() views must stay valid
        self.image_queue = [None] * self.maxsize
        self.data_source_queue = [None] * self.maxsize
        self.current_tail, self.full = 0, False

    def enqueue(self, items, item_images=None, data_source=None):
        if item_images is not None:
            assert len(items) == len(item_images)
        items = items.detach()
        if items.dtype != self.dtype:
            items = items.to(self.dtype)
        items = items.contiguous()
        old_tail, n = self.current_tail, items.shape[0]
        self.current_tail, self.full = ops.queue_enqueue(self.vector_queue, items, self.current_tail, self.full)
        # python-side bookkeeping of the parallel lists follows the same segments
        from .queue_index import enqueue_segments
        for dst, src, ln in enqueue_segments(old_tail, n, self.maxsize)[0]:
            if self.keep_images and item_images is not None:
                self.image_queue[dst:dst + ln] = list(item_images[src:src + ln])
            self.data_source_queue[dst:dst + ln] = [data_source] * ln

    def dequeue(self):
        # views, not copies (storage_queue.py:51-56): the next enqueue shows through them
        return dict(queue_vectors=self.vector_queue.detach(), queue_images=self.image_queue, queue_data_sources=self.data_source_queue)
