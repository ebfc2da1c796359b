"""Where the cloud's large arrays live in HBM (round 3; DESIGN.md section 3, measurements in ``profiles/r03_placement_*.txt``,
``r03_zone_*.txt``).

The densify kernel is bound by its row stores: ``points`` and ``normals``, 12 bytes per point each, the same row of both at the
same moment (the two list appends of ``scripts/test.py:238-240``, fused).  On MI355X the 288 GB of HBM fall into three classes
of physical address ranges of about a third of the memory each; two lock-step store streams inside ONE class run at 5.8 TB/s,
in two classes at 7.1-7.2 TB/s -- 2.94 vs 2.60 ms for the 185-view 1080p workload.  A fresh process is handed memory from one
end of the device, i.e. from one class: the "box state" lottery of rounds 1 and 2.

``ZoneArena`` (``csrc/ddarena.hip`` behind ``dd_arena_*``) takes physical memory in 1 GiB chunks through the virtual-memory API,
classifies every chunk with a two-stream store probe against one anchor chunk per class, and maps each requested array from
chunks of the classes its layout names: ``rotated(phase)`` -- chunk k from class (phase + k) mod 3 -- or class-pure per group.
``place_outputs`` is what ``CloudBuilder`` calls: points / normals / colours rotated with phases 0 / 1 / 2, so rows written in lock
step never share a class.  Nothing here changes a result: it only chooses physical pages.

Arrays from the arena are ordinary device tensors for every kernel, but their memory is not IPC-exportable: buffers that
RCCL sends or receives (``distributed.fuse_replicated``) are allocated normally and handed to ``CloudBuilder(buffers=...)``.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import lib

GIB = 1 << 30
MIN_ROWS = 128 << 20           # by default clouds below 128 Mi rows (1.5 GiB of points) are left where they land: finding the three
                               # classes costs 27 ms per GiB of memory looked at (0.05-2.7 s), more than small clouds can win back
GROUP_POINTS, GROUP_NORMALS, GROUP_OTHER = 0, 1, 2        # class-pure layouts: arrays of different groups in different classes


def rotated(phase: int = 0) -> int:
    """Layout code of an array whose consecutive chunks come from classes ``(phase + k) mod 3``."""
    return _lib.DD_ARENA_ROTATED + int(phase) % 3


def blocked() -> int:
    """Layout code of ONE large row array whose first, middle and last third come from three different classes (a cloud of
    points only: its scatter pass then takes tiles of the three thirds in turn -- ``CloudBuilder`` does that by itself)."""
    return _lib.DD_ARENA_BLOCKED


def default_layout() -> str:
    """``DD_PLACEMENT_LAYOUT`` = ``rotated`` (default: chunk k of every array from class (phase + k) mod 3, the phases of
    points / normals / colours differ) or ``separated`` (points, normals and colours class-pure in three different classes)."""
    m = os.environ.get("DD_PLACEMENT_LAYOUT", "rotated").lower()
    return m if m in ("rotated", "separated") else "rotated"


def default_mode() -> str:
    """``DD_PLACEMENT`` = ``probed`` (default) or ``first`` (take what the allocator returns, rounds 1-2 behaviour)."""
    m = os.environ.get("DD_PLACEMENT", "probed").lower()
    return m if m in ("probed", "first") else "probed"


@dataclass
class PlacementReport:
    """What ``place_outputs`` did -- kept on the ``CloudBuilder`` as ``.placement`` and printed by ``bench.py``."""
    mode: str                                   # "probed" | "first" | "skipped: <why>" | "degraded: <why>"
    classes: Optional[Dict[str, list]] = None   # per array: the classes of its chunks
    seconds: float = 0.0
    stats: Optional[dict] = None
    layout: Optional[str] = None                # "rotated" | "separated" | "blocked"

    def as_dict(self) -> dict:
        return {"mode": self.mode, "layout": self.layout, "classes": self.classes, "seconds": round(self.seconds, 3), "arena": self.stats}


class ArenaError(RuntimeError):
    pass


def _acheck(rc: int) -> int:
    if rc < 0:
        raise ArenaError(f"libddcore arena error {rc}: {lib.dd_arena_last_error().decode('utf-8', 'replace')}")
    return rc


class _Block:
    """One array of the arena, exported through ``__cuda_array_interface__``; torch keeps this object alive for as long as
    any tensor (or view) shares the memory, and the array goes back to the driver when the last of them dies."""

    def __init__(self, arena: "ZoneArena", ptr: int, nbytes: int):
        self._arena, self.ptr, self.nbytes = arena, ptr, nbytes

    @property
    def __cuda_array_interface__(self) -> dict:
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2, "strides": None}

    def __del__(self):
        arena = getattr(self, "_arena", None)
        if arena is not None and arena._handle is not None and self.ptr:
            try:
                lib.dd_arena_free(arena._handle, C.c_void_p(self.ptr))
            except Exception:      # noqa: BLE001  (interpreter shutdown)
                pass
            self.ptr = 0


class ZoneArena:
    """Process-wide arena of one device (``get_arena``).  ``alloc`` takes ``{name: (shape, dtype, group)}`` and returns
    ``{name: tensor}``; all arrays of one call are placed against each other."""

    def __init__(self, device: torch.device, chunk_bytes: int = 0):
        self.device = torch.device(device)
        h = C.c_void_p()
        _acheck(lib.dd_arena_create(self.device.index or 0, int(chunk_bytes), C.byref(h)))
        self._handle = h
        self._lock = threading.Lock()

    def alloc(self, specs: Dict[str, Tuple[Sequence[int], torch.dtype, int]], max_scout_bytes: Optional[int] = None) -> Tuple[Dict[str, torch.Tensor], bool]:
        """-> (tensors, degraded).  ``degraded``: some group had to share a class (budget or memory too small)."""
        names = list(specs)
        n = len(names)
        nbytes = [int(np.prod(specs[k][0], dtype=np.int64)) * torch.empty((), dtype=specs[k][1]).element_size() for k in names]
        if any(b <= 0 for b in nbytes):
            raise ValueError("arena arrays must not be empty")
        st = self.stats()
        chunk = max(int(st["chunk_bytes"]), 1)
        if sum(-(-b // chunk) for b in nbytes) > sum(st["chunks_pooled"]):
            # physical memory will be needed: blocks torch has freed but keeps cached are memory the driver cannot hand to the arena
            # (a request the arena's spare chunks and cached arrays cover -- a cloud per scene -- leaves torch's cache alone)
            torch.cuda.empty_cache()
        if max_scout_bytes is None:
            # a class is a third of the memory and an idle device hands out long stretches of one class (64 GiB seen): finding
            # all three may take a look at ~100 GiB.  Everything scouted and not used is back with the driver when alloc returns.
            free = torch.cuda.mem_get_info(self.device)[0]
            max_scout_bytes = int(min(128 * GIB, 0.8 * max(0, free - sum(nbytes))))
        sizes = (C.c_int64 * n)(*nbytes)
        groups = (C.c_int32 * n)(*[int(specs[k][2]) for k in names])
        ptrs = (C.c_void_p * n)()
        with self._lock:
            rc = _acheck(lib.dd_arena_alloc(self._handle, n, sizes, groups, int(max_scout_bytes), ptrs))
        out = {}
        for k, b, p in zip(names, nbytes, ptrs):
            shape, dtype, _ = specs[k]
            raw = torch.as_tensor(_Block(self, int(p), b), device=self.device)
            out[k] = raw.view(dtype).view(tuple(shape))
        return out, rc == 1

    def classes_of(self, t: torch.Tensor) -> list:
        """Class of every chunk behind an arena array (``t`` must start at the array's first byte)."""
        buf = (C.c_int32 * 512)()
        k = _acheck(lib.dd_arena_classes(self._handle, C.c_void_p(t.data_ptr()), buf, 512))
        return [int(buf[i]) for i in range(min(k, 512))]

    def probe_ms(self, a: torch.Tensor, b: torch.Tensor) -> float:
        """The two-stream store probe on the first ``stats()['probe_bytes']`` bytes of ``a`` and ``b`` (OVERWRITES them)."""
        need = self.stats()["probe_bytes"]
        if a.numel() * a.element_size() < need or b.numel() * b.element_size() < need:
            raise ValueError(f"probe windows need {need} bytes")
        ms = C.c_float()
        _acheck(lib.dd_arena_probe(self._handle, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.byref(ms)))
        return float(ms.value)

    def trim(self, pool_chunks_per_class: int = -1) -> None:
        """Give the spare classified chunks back to the driver (``pool_chunks_per_class`` >= 0: and keep that many per class
        from now on; the default is 4, i.e. up to 12 GiB idle between allocations)."""
        _acheck(lib.dd_arena_trim(self._handle, int(pool_chunks_per_class)))

    def stats(self) -> dict:
        s = _lib.DDArenaStats()
        _acheck(lib.dd_arena_stats(self._handle, C.byref(s)))
        return {"chunk_bytes": int(s.chunk_bytes), "probe_bytes": int(s.probe_bytes), "num_classes": int(s.num_classes),
                "degraded_allocs": int(s.degraded_allocs), "chunks_created": int(s.chunks_created), "chunks_released": int(s.chunks_released),
                "probes": int(s.probes), "chunks_held": [int(x) for x in s.chunks_held], "chunks_pooled": [int(x) for x in s.chunks_pooled],
                "same_class_ms": round(float(s.same_class_ms), 4),
                "cross_class_ms": round(float(s.cross_class_ms), 4), "seconds": round(float(s.seconds), 3)}


_arenas: Dict[int, ZoneArena] = {}
_arenas_lock = threading.Lock()


def get_arena(device) -> ZoneArena:
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _arenas_lock:
        a = _arenas.get(idx)
        if a is None:
            a = _arenas[idx] = ZoneArena(torch.device("cuda", idx), int(os.environ.get("DD_ARENA_CHUNK_MIB", "0")) << 20)      # 0 = the library's default
        return a


def keep_everything(device) -> None:
    """From now on the device's arena gives NO chunk back to the driver: freed chunks stay in its pool, classified, and serve the next
    request.  For a process under ``rocprofv3``: with the profiler loaded ``hipMemRelease`` does not return the memory to the device
    (measured: ``tools/arena_free_probe.py``, ``profiles/r05_arena_free_under_rocprofv3.txt``), so a chunk that is released is lost
    until the process ends, while a pooled one is used again."""
    _acheck(lib.dd_arena_set_pool(get_arena(device)._handle, 1 << 20))      # (not trim(): that would release every spare chunk and the mapped cache right now)


def trim(device=None) -> None:
    """Give the spare chunks of the device's arena (if one exists) back to the driver, e.g. before a large plain allocation."""
    with _arenas_lock:
        arenas = list(_arenas.values()) if device is None else [a for i, a in _arenas.items() if i == (torch.device(device).index or 0)]
    for a in arenas:
        a.trim()


def place_arrays(specs: Dict[str, Tuple[Sequence[int], torch.dtype, int]], device, mode: Optional[str] = None) -> Tuple[Dict[str, torch.Tensor], PlacementReport]:
    """Allocate the arrays of ``specs`` = ``{name: (shape, dtype, group)}``: through the arena (``probed``), or plainly
    (``first``, and whenever the arena cannot serve the request -- the report says why)."""
    import time
    mode = mode or default_mode()
    dev = torch.device(device)
    plain = lambda: {k: torch.empty(tuple(s), dtype=d, device=dev) for k, (s, d, _) in specs.items()}
    if mode == "first":
        return plain(), PlacementReport("first")
    t0 = time.perf_counter()
    try:
        arena = get_arena(dev)
        tensors, degraded = arena.alloc(specs)
    except (ArenaError, RuntimeError) as e:          # the virtual-memory API is missing or out of memory: carry on unplaced
        free, total = torch.cuda.mem_get_info(dev)
        want = sum(int(np.prod(sh, dtype=np.int64)) * torch.empty((), dtype=dt).element_size() for sh, dt, _ in specs.values())
        why = f"skipped: {e} (request {want / GIB:.1f} GiB; {free / GIB:.1f} of {total / GIB:.1f} GiB free afterwards, torch holds {torch.cuda.memory_reserved(dev) / GIB:.1f} GiB)"
        return plain(), PlacementReport(why[:300])
    rep = PlacementReport("degraded: some arrays share a class" if degraded else "probed", seconds=time.perf_counter() - t0)
    rep.classes = {k: _summary(arena.classes_of(t)) for k, t in tensors.items()}
    rep.stats = arena.stats()
    return tensors, rep


def _spares_cover(specs: Dict[str, Tuple[Sequence[int], torch.dtype, int]], device) -> bool:
    """True when the device's arena already exists and its pool of classified spare chunks can serve ``specs`` with the two
    lock-step arrays (points, normals) in different classes -- placing the cloud then costs no scouting at all, whatever its size
    (round 4: a 60-view scan or a scene of a batch after the first takes a placed cloud too)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _arenas_lock:
        arena = _arenas.get(idx)
    if arena is None:
        return False
    try:
        st = arena.stats()
    except ArenaError:
        return False
    chunk = max(int(st["chunk_bytes"]), 1)
    need = sorted((-(-int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dt).element_size() // chunk) for shape, dt, _ in specs.values()), reverse=True)
    have = sorted(st["chunks_pooled"], reverse=True)
    # the largest arrays (points, normals) each from a class of their own, the rest (colours) from whatever is left
    if len(need) >= 2:
        return have[0] >= need[0] and have[1] >= need[1] and sum(have) >= sum(need)
    return sum(have) >= sum(need)


def _summary(classes: list) -> list:
    """Class sequence of an array's chunks, condensed: [0, 1, 2, 0] stays, long ones become a description."""
    if len(classes) <= 6:
        return classes
    if all(classes[k] == (classes[0] + k) % 3 for k in range(len(classes))):
        return [f"{len(classes)} chunks rotating from class {classes[0]}"]
    runs = [[classes[0], 1]]
    for c in classes[1:]:
        if c == runs[-1][0]:
            runs[-1][1] += 1
        else:
            runs.append([c, 1])
    if len(runs) <= 4:
        return [" + ".join(f"{k} chunks of class {c}" for c, k in runs)]
    if len(set(classes)) == 1:
        return [f"{len(classes)} chunks of class {classes[0]}"]
    return classes


def place_outputs(capacity: int, *, colors: bool, device, mode: Optional[str] = None, normals: bool = True,
                  layout: Optional[str] = None) -> tuple:
    """(points, normals | None, colors | None, PlacementReport) for the row arrays of a cloud of ``capacity`` rows."""
    n = max(int(capacity), 1)
    layout = layout or default_layout()
    codes = (GROUP_POINTS, GROUP_NORMALS, GROUP_OTHER) if layout == "separated" else (rotated(0), rotated(1), rotated(2))
    if not normals:
        # one row stream only: nothing to keep apart at equal rows.  Its thirds go to the three classes instead, and the
        # scatter pass walks the thirds in turn (DDViewBatch.tuning bits 8-13; CloudBuilder.append decides)
        layout, codes = "blocked", (blocked(), GROUP_NORMALS, GROUP_OTHER)
    specs = {"points": ((n, 3), torch.float32, codes[0])}
    if normals:
        specs["normals"] = ((n, 3), torch.float32, codes[1])
    if colors:
        specs["colors"] = ((n, 3), torch.uint8, codes[2])
    explicit = mode is not None
    mode = mode or default_mode()
    if mode != "first" and n < MIN_ROWS and not explicit and not _spares_cover(specs, device):
        t, rep = place_arrays(specs, device, "first")
        rep.mode = (f"skipped: {n} rows < {MIN_ROWS} and the arena holds no classified spare chunks for them (scouting the memory costs "
                    "more than a cloud this small wins back; placement='probed' forces it)")
    else:
        t, rep = place_arrays(specs, device, mode)
    if rep.mode.startswith(("probed", "degraded")):      # (degraded: some chunks lie outside the class asked for -- the layout still holds mostly)
        rep.layout = layout
    return t["points"], t.get("normals"), t.get("colors"), rep
