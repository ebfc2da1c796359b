#!/usr/bin/env python3
"""Interleaved A/B of BUILD variants of the densify kernels in ONE process (GPU box): every variant is its own
libddcore_<tag>.so (compiled here with extra -D flags), all are loaded side by side with ctypes and timed round-robin on
the same resident buffers -- process-to-process and box-to-box spread (several %) cannot leak into the comparison.

    python tools/ab_builds.py --workload garden185 --views 96 base: sp8:-DDD_SP_WAVES=8 sp16:-DDD_SP_WAVES=16
"""
import argparse
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import depthdensifier_amd as dd  # noqa: E402
from depthdensifier_amd import _lib  # noqa: E402


def build(tag: str, flags: list) -> C.CDLL:
    out = ROOT / "build" / "ab" / f"libddcore_{tag}.so"
    out.parent.mkdir(parents=True, exist_ok=True)
    src = ROOT / "depthdensifier_amd" / "csrc" / "ddcore.hip"
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{ROOT / 'include'}", str(src), "-o", str(out)] + flags
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lib = C.CDLL(str(out))
    lib.dd_unproject_compact.restype = C.c_int
    lib.dd_unproject_compact.argtypes = [C.POINTER(_lib.DDViewBatch), C.POINTER(_lib.DDCloudOut), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="garden185")
    ap.add_argument("--views", type=int, default=96)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--mask-kind", default="blob")
    ap.add_argument("--tuning", type=int, default=0, help="DDViewBatch.tuning for every variant (4 = two-pass)")
    ap.add_argument("variants", nargs="+", help="tag:flag,flag,...  (empty flag list = the committed defaults)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = dict(bench.WORKLOADS[a.workload]); cfg["V"] = a.views; cfg["mask_kind"] = a.mask_kind
    ids = np.arange(a.views)
    scene = bench.make_scene(cfg, ids, dev)
    H, W = cfg["H"], cfg["W"]
    params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (a.views, 1))
    batch = dd.ViewBatch(scene["depth"], params, bench.ring_poses(ids, a.views), mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"],
                         conf=scene["conf"], conf_threshold=cfg.get("conf"), device=dev, tuning=a.tuning)
    builder = dd.CloudBuilder(batch.max_points, normals=cfg["normal"], colors=cfg["rgb"], pixel_index=False, device=dev)
    cb, out = batch.c_struct(), builder._out_struct()
    ws = torch.zeros(4 * batch.workspace_bytes() + 4096, dtype=torch.uint8, device=dev)      # variants with smaller tiles need more
    offs = torch.empty(a.views + 1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    libs = []
    for spec in a.variants:
        tag, _, fl = spec.partition(":")
        libs.append((tag, build(tag, [f for f in fl.split(",") if f]), []))
    ref = None
    for r in range(a.rounds + 1):
        for tag, lib, times in libs:
            builder.cursor.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.dd_unproject_compact(C.byref(cb), C.byref(out), offs.data_ptr(), builder.cursor.data_ptr(), ws.data_ptr(), ws.numel(), stream)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0, (tag, rc)
            if r == 0:
                chk = (int(builder.cursor.item()), float(builder.xyz[: int(builder.cursor.item())].double().sum()))
                ref = ref or chk
                assert chk == ref, f"variant {tag} computes something else: {chk} vs {ref}"
            else:
                times.append(e0.elapsed_time(e1))
    n = int(builder.cursor.item())
    alg = bench.algorithmic_bytes(cfg, a.views, n, False)
    print(f"{a.workload} x {a.views} views, {n} points, algorithmic {alg / 1e9:.3f} GB")
    base = float(np.median(libs[0][2]))
    for tag, _, times in libs:
        med = float(np.median(times))
        print(f"{tag:14s} median {med:8.3f} ms  min {min(times):8.3f}  frac {alg / med / 1e6 / 8000:6.3f}  vs first {base / med:6.3f}x")


if __name__ == "__main__":
    main()
