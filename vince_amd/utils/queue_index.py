"""Pure-integer ring-buffer arithmetic of StorageQueue.enqueue (reference utils/storage_queue.py:31-49).

Host-side mirror of the C implementation in csrc/misc.hip (vince_queue_enqueue); used for the python-side parallel
lists and by the data-parallel glue.  CPU-testable.

Written as CLOSED-FORM modular arithmetic -- the reference recurses (storage_queue.py:43) and the oracle restates that as a loop
(oracle/vince_oracle.py enqueue_segments); the two formulations are independent and both pinned to golden set G1."""


def enqueue_segments(tail, n, maxsize):
    """Returns ([(dst_start, src_start, length), ...] in execution order, new_tail, wrapped).

    `tail` is in [0, maxsize] (an exact fit leaves it AT maxsize; only the next write wraps).  Rows that fit go in one piece.
    Otherwise the first piece fills the ring up to its end, then come `laps` whole passes over the ring (n > maxsize: later rows
    overwrite earlier ones, as the reference's recursion does) and a last piece of 1 .. maxsize rows starting at slot 0 -- a
    remainder of exactly maxsize rows is one full pass that leaves the tail AT maxsize, hence the (rest - 1)."""
    room = maxsize - tail
    if n <= room:
        return ([(tail, 0, n)] if n > 0 else []), tail + n, False
    rest = n - room                            # >= 1 rows still to place, all starting at slot 0
    laps = (rest - 1) // maxsize
    last = rest - laps * maxsize               # 1 .. maxsize
    segs = [(tail, 0, room)] if room > 0 else []
    segs += [(0, room + i * maxsize, maxsize) for i in range(laps)]
    segs.append((0, room + laps * maxsize, last))
    return segs, last, True
