"""Streaming writer of the dense points into ``points3D.bin``.

The reference adds every dense point with ``rec.add_point3D(xyz, Track(), color)`` in a Python loop
(``scripts/test.py:355-358``) and then serialises the whole reconstruction (``:363``).  At full density the
file is the largest object of a run -- 51 bytes per point, 16.6 GB for a 185-view 1080p scan, 170 GB for the
2000-view scene -- and building it on the host needs the float64 copies plus the record array in memory at
once.  Here the records are formatted on the GPU (``dd_format_points3d``: id, float64 xyz, rgb, error -1, empty
track, in COLMAP's byte layout) chunk by chunk; the host only moves bytes: two pinned buffers, the device->host
copy of chunk k+1 overlapping the file write of chunk k.  Host memory: two chunks, whatever the cloud's size.

``write_dense_records(f, cloud, first_id)`` appends to an open file; ``write_dense_at(path, offset, ...)`` writes at
a byte offset of an existing file (each rank of a multi-GPU run writes its own slice -- no gather at all).
"""

from __future__ import annotations

import ctypes as C
import os
import queue
import threading
from typing import Optional

import torch

from ._lib import DDCoreError, lib

RECORD_BYTES = 51                     # uint64 id | 3 x float64 | 3 x uint8 | float64 error | uint64 track length
DEFAULT_CHUNK_POINTS = 4 << 20        # 204 MiB of records per chunk


def _fields(cloud):
    """(xyz ptr, rgb ptr, packed ptr, keep-alive tensors) of a device cloud: the 16-byte records when it has them."""
    if getattr(cloud, "packed", None) is not None:
        t = cloud.packed if cloud.packed.is_contiguous() else cloud.packed.contiguous()
        return None, None, t, (t,)
    pts = cloud.points if cloud.points.is_contiguous() else cloud.points.contiguous()
    if pts.dtype != torch.float32:
        pts = pts.to(torch.float32)
    col = cloud.colors
    if col is not None and not col.is_contiguous():
        col = col.contiguous()
    return pts, col, None, (pts, col)


class _Sink:
    """Where the bytes go: sequential ``write`` on an open file, or ``pwrite`` at a moving offset of a descriptor."""

    def __init__(self, f=None, fd: Optional[int] = None, offset: int = 0):
        self.f, self.fd, self.offset = f, fd, offset

    def put(self, view) -> None:
        if self.f is not None:
            self.f.write(view)
            return
        done = 0
        while done < len(view):
            done += os.pwrite(self.fd, view[done:], self.offset + done)
        self.offset += len(view)


def _stream(sink: _Sink, cloud, first_id: int, lo: int, hi: int, chunk_points: int) -> int:
    n = hi - lo
    if n <= 0:
        return 0
    xyz, rgb, packed, keep = _fields(cloud)
    dev = keep[0].device
    if dev.type != "cuda":
        raise RuntimeError("the streaming model writer formats the records on the GPU (no CPU fallback)")
    chunk = max(1, min(int(chunk_points), n))
    nb = chunk * RECORD_BYTES
    dbuf = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(2)]
    hbuf = [torch.empty(nb, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    hview = [memoryview(h.numpy()) for h in hbuf]
    main = torch.cuda.current_stream(dev)
    copy = torch.cuda.Stream(dev)
    free = [threading.Semaphore(1), threading.Semaphore(1)]     # a slot is free once its bytes are in the file
    jobs: "queue.Queue" = queue.Queue()
    failure: list = []

    def writer():
        while True:
            job = jobs.get()
            if job is None:
                return
            slot, nbytes, done = job
            try:
                if not failure:
                    done.synchronize()
                    sink.put(hview[slot][:nbytes])
            except BaseException as e:      # noqa: BLE001  (reported by the main thread)
                failure.append(e)
            finally:
                free[slot].release()

    ptr = lambda t, row, width: None if t is None else t.data_ptr() + row * width * t.element_size()
    th = threading.Thread(target=writer, name="dd-model-writer", daemon=True)
    th.start()
    try:
        for k, a in enumerate(range(lo, hi, chunk)):
            b = min(a + chunk, hi)
            slot = k & 1
            free[slot].acquire()
            if failure:
                free[slot].release()
                break
            rc = lib.dd_format_points3d(ptr(xyz, a, 3), ptr(rgb, a, 3), ptr(packed, a, 4), b - a, C.c_uint64(first_id + (a - lo)),
                                        dbuf[slot].data_ptr(), main.cuda_stream)
            if rc < 0:
                free[slot].release()
                raise DDCoreError(rc, lib.dd_model_last_error().decode())
            ready = torch.cuda.Event()
            ready.record(main)
            copy.wait_event(ready)
            nbytes = (b - a) * RECORD_BYTES
            with torch.cuda.stream(copy):
                hbuf[slot][:nbytes].copy_(dbuf[slot][:nbytes], non_blocking=True)
                done = torch.cuda.Event()
                done.record(copy)
            jobs.put((slot, nbytes, done))
    finally:
        jobs.put(None)
        th.join()
    if failure:
        raise failure[0]
    main.wait_stream(copy)                # the device buffers may be reused by later work on the current stream
    return n


def write_dense_records(f, cloud, first_id: int, chunk_points: int = DEFAULT_CHUNK_POINTS) -> int:
    """Append the ``len(cloud)`` dense points of a device cloud to the open binary file ``f`` as points3D.bin records
    with ids ``first_id, first_id + 1, ...`` (``scripts/test.py:355-358`` + ``:363`` for the dense points)."""
    return _stream(_Sink(f=f), cloud, int(first_id), 0, len(cloud), chunk_points)


def write_dense_at(path, byte_offset: int, cloud, first_id: int, chunk_points: int = DEFAULT_CHUNK_POINTS) -> int:
    """Write the records of a device cloud at ``byte_offset`` of the existing file ``path`` (``pwrite``): the sharded
    model write -- every rank of a multi-GPU run puts its own slice of the dense points where the plan says, so the
    clouds are never gathered.  ``first_id``: the id of this cloud's first point."""
    fd = os.open(str(path), os.O_WRONLY)
    try:
        return _stream(_Sink(fd=fd, offset=int(byte_offset)), cloud, int(first_id), 0, len(cloud), chunk_points)
    finally:
        os.close(fd)
