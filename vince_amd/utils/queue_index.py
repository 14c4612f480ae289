"""Pure-integer ring-buffer arithmetic of StorageQueue.enqueue (reference utils/storage_queue.py:31-49).

Host-side mirror of the C implementation in csrc/misc.hip (vince_queue_enqueue); used for the python-side parallel
lists and by the data-parallel glue.  CPU-testable."""


def enqueue_segments(tail, n, maxsize):
    """Returns ([(dst_start, src_start, length), ...] in execution order, new_tail, wrapped)."""
    segs = []
    src = 0
    wrapped = False
    while True:
        if tail + n > maxsize:
            num_start = maxsize - tail
            if num_start > 0:
                segs.append((tail, src, num_start))
            tail = 0
            wrapped = True
            src += num_start
            n -= num_start
        else:
            if n > 0:
                segs.append((tail, src, n))
            return segs, tail + n, wrapped
