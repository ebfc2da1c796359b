"""ctypes binding of ``libddcore.so`` (C ABI declared in ``include/ddcore.h``).

There is no CPU fallback: if the HIP library has not been built, importing this
module raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C depthdensifier_amd/csrc``.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# torch MUST be imported before libddcore.so is dlopen'ed: libtorch_hip.so asks for its bundled
# "libamdhip64.so" by file name while libddcore.so asks for the soname "libamdhip64.so.7".  With
# torch first the loader satisfies our soname from torch's already-loaded runtime (one HIP
# runtime, shared streams); the other order loads two runtimes and every stream handle torch
# passes us is foreign (hipMemsetAsync fails with an invalid handle).
import torch  # noqa: F401  (load order, see above)

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("DDCORE_LIB", _HERE / "libddcore.so"))

DD_ABI_VERSION = 14
DD_OK = 0
DD_F32, DD_F16 = 0, 1
DD_NPY_F32, DD_NPY_F16, DD_NPY_U8, DD_NPY_BOOL = 0, 1, 2, 3
DD_VALID_DEPTH_POSITIVE = 0x1
DD_VALID_MASK = 0x2
DD_VALID_CONF = 0x4
DD_ROTATE_NORMALS = 0x8
DD_REFINE = 0x10

#: every symbol include/ddcore.h declares (and, last, the two of include/ddcore_lab.h: the experiment switches of tests and A/B tools)
EXPORTS = (
    "dd_abi_version",
    "dd_last_error",
    "dd_count_valid",
    "dd_workspace_bytes",
    "dd_plan",
    "dd_scatter",
    "dd_unproject_compact",
    "dd_stream_fork",
    "dd_stream_wait",
    "dd_streams_overlap",
    "dd_chain_workgroup_limit",
    "dd_floater_votes",
    "dd_filter_last_error",
    "dd_votes_workspace_bytes",
    "dd_compact_workspace_bytes",
    "dd_compact_cloud",
    "dd_refine_apply",
    "dd_refine_last_error",
    "dd_refine_fit",
    "dd_refine_fit_async",
    "dd_sort_knots",
    "dd_allgatherv",
    "dd_comm_last_error",
    "dd_npy_header",
    "dd_npy_read",
    "dd_upload_async",
    "dd_ingest_last_error",
    "dd_prefetch_create",
    "dd_prefetch_submit",
    "dd_prefetch_wait",
    "dd_prefetch_release",
    "dd_prefetch_destroy",
    "dd_format_points3d",
    "dd_model_last_error",
    "dd_arena_create",
    "dd_arena_alloc",
    "dd_arena_free",
    "dd_arena_trim",
    "dd_arena_set_pool",
    "dd_arena_classes",
    "dd_arena_probe",
    "dd_arena_stats",
    "dd_arena_destroy",
    "dd_arena_last_error",
    "dd_debug_tuning",
    "dd_debug_plan",
)
LAB_EXPORTS = ("dd_debug_tuning", "dd_debug_plan")


class DDViewBatch(C.Structure):
    _fields_ = [
        ("num_views", C.c_int32),
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("stride", C.c_int32),
        ("depth", C.c_void_p),
        ("mask", C.c_void_p),
        ("conf", C.c_void_p),
        ("normal", C.c_void_p),
        ("rgb", C.c_void_p),
        ("params", C.c_void_p),
        ("depth_dtype", C.c_int32),
        ("conf_dtype", C.c_int32),
        ("conf_threshold", C.c_float),
        ("flags", C.c_uint32),
        ("view_index_base", C.c_int32),
        ("tuning", C.c_uint32),
        ("refined_out", C.c_void_p),
        ("chain", C.c_void_p),
        ("chain_seq", C.c_int64),
    ]


class DDCloudOut(C.Structure):
    _fields_ = [
        ("xyz", C.c_void_p),
        ("normal", C.c_void_p),
        ("rgb", C.c_void_p),
        ("pixel_index", C.c_void_p),
        ("view_index", C.c_void_p),
        ("capacity", C.c_int64),
        ("xyz_rgba", C.c_void_p),
    ]


class DDFilterViews(C.Structure):
    _fields_ = [
        ("num_views", C.c_int32),
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("reserved", C.c_int32),
        ("depth", C.c_void_p),
        ("mask", C.c_void_p),
        ("cams", C.c_void_p),
        ("grazing_cos", C.c_double),
        ("depth_threshold", C.c_float),
        ("reserved2", C.c_float),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
        ("mode", C.c_int32),
        ("reserved3", C.c_int32),
    ]


DD_ARENA_ROTATED = 8
DD_ARENA_BLOCKED = 16
# DDViewBatch.tuning (include/ddcore.h): what a caller may choose
DD_TUNE_GENERIC, DD_TUNE_TWO_PASS, DD_TUNE_SINGLE_PASS, DD_TUNE_DENSE_TILES = 1, 4, 8, 128
DD_TUNE_INTERLEAVE_MASK = 63 << 8
DD_TUNE_ASSUME_DENSE = 1 << 17      # count-free plan, verified by the scatter pass
DD_TUNE_TILE_SMALL, DD_TUNE_TILE_LARGE = 1 << 18, 3 << 18
DD_TUNE_BY_INDEX = 1 << 22          # single-pass tiles taken by workgroup index, not by ticket (the caller has the GPU to itself)
DD_TUNE_ALL = (DD_TUNE_GENERIC | DD_TUNE_TWO_PASS | DD_TUNE_SINGLE_PASS | DD_TUNE_DENSE_TILES | DD_TUNE_INTERLEAVE_MASK | DD_TUNE_ASSUME_DENSE
               | DD_TUNE_TILE_LARGE | DD_TUNE_BY_INDEX)


def DD_TUNE_INTERLEAVE(k: int) -> int:
    return ((int(k) - 1) & 63) << 8


# include/ddcore_lab.h: the thread-local experiment switches (tests, A/B tools) -- never set by the product path
DD_LAB_LIST_ORDER, DD_LAB_FAULT_INJECT, DD_LAB_POLL_LANES_32, DD_LAB_POLL_LANES_64 = 1, 2, 4, 8
DD_LAB_LOOKBACK, DD_LAB_APPLY_PLAIN = 16, 128


def DD_LAB_APPLY_WGS(n: int) -> int:
    return (int(n) & 0x1FFF) << 8


class DDArenaStats(C.Structure):
    _fields_ = [
        ("chunk_bytes", C.c_int64),
        ("probe_bytes", C.c_int64),
        ("num_classes", C.c_int32),
        ("degraded_allocs", C.c_int32),
        ("chunks_created", C.c_int64),
        ("chunks_released", C.c_int64),
        ("probes", C.c_int64),
        ("chunks_held", C.c_int64 * 3),
        ("chunks_pooled", C.c_int64 * 3),
        ("same_class_ms", C.c_float),
        ("cross_class_ms", C.c_float),
        ("seconds", C.c_double),
    ]


class DDCoreError(RuntimeError):
    """A negative return code from libddcore.so."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libddcore error {code}: {message}")
        self.code = code


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: the HIP core is not built. There is no CPU fallback; "
            "run `python -c \"import __graft_entry__ as g; g.build()\"` at the repo root."
        )
    lib = C.CDLL(str(LIB_PATH))
    lib.dd_abi_version.restype = C.c_int
    lib.dd_abi_version.argtypes = []
    lib.dd_last_error.restype = C.c_char_p
    lib.dd_last_error.argtypes = []
    lib.dd_count_valid.restype = C.c_int
    lib.dd_count_valid.argtypes = [C.POINTER(DDViewBatch), C.c_void_p, C.c_void_p]
    lib.dd_workspace_bytes.restype = C.c_int64
    lib.dd_workspace_bytes.argtypes = [C.POINTER(DDViewBatch)]
    lib.dd_plan.restype = C.c_int
    lib.dd_plan.argtypes = [C.POINTER(DDViewBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.dd_scatter.restype = C.c_int
    lib.dd_scatter.argtypes = [C.POINTER(DDViewBatch), C.POINTER(DDCloudOut), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.dd_unproject_compact.restype = C.c_int
    lib.dd_unproject_compact.argtypes = [
        C.POINTER(DDViewBatch), C.POINTER(DDCloudOut), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
    ]
    lib.dd_stream_fork.restype = C.c_int
    lib.dd_stream_fork.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dd_stream_wait.restype = C.c_int
    lib.dd_stream_wait.argtypes = [C.c_void_p, C.c_void_p]
    lib.dd_streams_overlap.restype = C.c_int
    lib.dd_streams_overlap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.dd_chain_workgroup_limit.restype = C.c_int32
    lib.dd_chain_workgroup_limit.argtypes = []
    lib.dd_debug_tuning.restype = C.c_uint32
    lib.dd_debug_tuning.argtypes = [C.c_uint32]
    lib.dd_debug_plan.restype = C.c_int
    lib.dd_debug_plan.argtypes = [C.POINTER(DDViewBatch), C.POINTER(C.c_int32)]
    lib.dd_floater_votes.restype = C.c_int
    lib.dd_floater_votes.argtypes = [C.POINTER(DDFilterViews), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
    lib.dd_votes_workspace_bytes.restype = C.c_int64
    lib.dd_votes_workspace_bytes.argtypes = [C.c_int32, C.c_int64]
    lib.dd_filter_last_error.restype = C.c_char_p
    lib.dd_filter_last_error.argtypes = []
    lib.dd_compact_workspace_bytes.restype = C.c_int64
    lib.dd_compact_workspace_bytes.argtypes = [C.c_int64]
    lib.dd_compact_cloud.restype = C.c_int
    lib.dd_compact_cloud.argtypes = [C.POINTER(DDCloudOut), C.c_int64, C.c_void_p, C.c_int32, C.POINTER(DDCloudOut), C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    lib.dd_refine_apply.restype = C.c_int
    lib.dd_refine_apply.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.dd_refine_last_error.restype = C.c_char_p
    lib.dd_refine_last_error.argtypes = []
    lib.dd_refine_fit.restype = C.c_int
    lib.dd_refine_fit.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dd_refine_fit_async.restype = C.c_int
    lib.dd_refine_fit_async.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dd_sort_knots.restype = C.c_int
    lib.dd_sort_knots.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dd_allgatherv.restype = C.c_int
    lib.dd_allgatherv.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(DDCloudOut), C.POINTER(C.c_int64), C.c_int32, C.c_void_p]
    lib.dd_comm_last_error.restype = C.c_char_p
    lib.dd_comm_last_error.argtypes = []
    lib.dd_npy_header.restype = C.c_int
    lib.dd_npy_header.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.dd_npy_read.restype = C.c_int
    lib.dd_npy_read.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_void_p, C.c_int64]
    lib.dd_upload_async.restype = C.c_int
    lib.dd_upload_async.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]
    lib.dd_prefetch_create.restype = C.c_int
    lib.dd_prefetch_create.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]
    lib.dd_prefetch_submit.restype = C.c_int64
    lib.dd_prefetch_submit.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.dd_prefetch_wait.restype = C.c_int
    lib.dd_prefetch_wait.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
    lib.dd_prefetch_release.restype = C.c_int
    lib.dd_prefetch_release.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.dd_prefetch_destroy.restype = C.c_int
    lib.dd_prefetch_destroy.argtypes = [C.c_void_p]
    lib.dd_ingest_last_error.restype = C.c_char_p
    lib.dd_ingest_last_error.argtypes = []
    lib.dd_format_points3d.restype = C.c_int
    lib.dd_format_points3d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.dd_model_last_error.restype = C.c_char_p
    lib.dd_model_last_error.argtypes = []
    lib.dd_arena_create.restype = C.c_int
    lib.dd_arena_create.argtypes = [C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]
    lib.dd_arena_alloc.restype = C.c_int
    lib.dd_arena_alloc.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_void_p)]
    lib.dd_arena_free.restype = C.c_int
    lib.dd_arena_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.dd_arena_trim.restype = C.c_int
    lib.dd_arena_trim.argtypes = [C.c_void_p, C.c_int32]
    lib.dd_arena_set_pool.restype = C.c_int
    lib.dd_arena_set_pool.argtypes = [C.c_void_p, C.c_int32]
    lib.dd_arena_classes.restype = C.c_int
    lib.dd_arena_classes.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
    lib.dd_arena_probe.restype = C.c_int
    lib.dd_arena_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    lib.dd_arena_stats.restype = C.c_int
    lib.dd_arena_stats.argtypes = [C.c_void_p, C.POINTER(DDArenaStats)]
    lib.dd_arena_destroy.restype = C.c_int
    lib.dd_arena_destroy.argtypes = [C.c_void_p]
    lib.dd_arena_last_error.restype = C.c_char_p
    lib.dd_arena_last_error.argtypes = []
    got = lib.dd_abi_version()
    if got != DD_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {got}, binding expects {DD_ABI_VERSION}; rebuild the library")
    return lib


lib = _load()


def check(rc: int) -> int:
    if rc < 0:
        raise DDCoreError(int(rc), lib.dd_last_error().decode("utf-8", "replace"))
    return rc


class lab_switches:
    """``with lab_switches(bits):`` -- the library calls this thread makes inside run with the experiment switches of
    ``include/ddcore_lab.h`` (``DD_LAB_*``); zero bits cost nothing.  Tests and A/B tools only."""

    def __init__(self, bits: int):
        self.bits = int(bits)

    def __enter__(self):
        if self.bits:
            self._before = lib.dd_debug_tuning(self.bits)
        return self

    def __exit__(self, *exc):
        if self.bits:
            lib.dd_debug_tuning(self._before)
        return False
