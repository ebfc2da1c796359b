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
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
import lab_bits  # noqa: E402
import depthdensifier_amd as dd  # noqa: E402
from depthdensifier_amd import _lib  # noqa: E402


BUILD_ONLY = False


def build(tag: str, flags: list) -> C.CDLL:
    out = ROOT / "build" / "ab" / f"libddcore_{tag}.so"
    out.parent.mkdir(parents=True, exist_ok=True)
    src = ROOT / "depthdensifier_amd" / "csrc" / "ddcore.hip"
    stamp = out.with_suffix(".flags")
    fresh = out.exists() and stamp.exists() and stamp.read_text() == " ".join(flags) and out.stat().st_mtime >= src.stat().st_mtime
    if not fresh:       # (variants built in the CPU container travel to the GPU box with the snapshot: build/ is not gpurun-ignored)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{ROOT / 'include'}", str(src), "-o", str(out)] + flags
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        stamp.write_text(" ".join(flags))
    if BUILD_ONLY:
        return None
    lib = C.CDLL(str(out))
    lib.dd_unproject_compact.restype = C.c_int
    lib.dd_debug_tuning.restype, lib.dd_debug_tuning.argtypes = C.c_uint32, [C.c_uint32]
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
    ap.add_argument("--stamps", action="store_true", help="variants built with -DDD_X_STAMPS: print the per-tile phase times (shader clocks)")
    ap.add_argument("--build-only", action="store_true", help="compile the variants (CPU container) and stop")
    a = ap.parse_args()
    if a.build_only:
        global BUILD_ONLY
        BUILD_ONLY = True
        for spec in a.variants:
            tag, _, fl = spec.partition(":")
            build(tag.partition("@")[0], [f for f in fl.split(",") if f])
            print("built", tag)
        return
    dev = torch.device("cuda", 0)
    cfg = dict(bench.WORKLOADS[a.workload]); cfg["V"] = a.views; cfg["mask_kind"] = a.mask_kind
    ids = np.arange(a.views)
    scene = bench.make_scene(cfg, ids, dev)
    H, W = cfg["H"], cfg["W"]
    params = np.tile([0.8 * W, 0.8 * W, W / 2.0, H / 2.0], (a.views, 1))
    batch = dd.ViewBatch(scene["depth"], params, bench.ring_poses(ids, a.views), mask=scene["mask"], normal=scene["normal"], rgb=scene["rgb"],
                         conf=scene["conf"], conf_threshold=cfg.get("conf"), device=dev, tuning=lab_bits.split(a.tuning)[0])
    builder = dd.CloudBuilder(batch.max_points, normals=cfg["normal"], colors=cfg["rgb"], pixel_index=False, device=dev)
    cb, out = batch.c_struct(), builder._out_struct()
    ws = torch.zeros(4 * batch.workspace_bytes() + 4096, dtype=torch.uint8, device=dev)      # variants with smaller tiles need more
    offs = torch.empty(a.views + 1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    libs = []
    tunings = {}
    for spec in a.variants:
        tag, _, fl = spec.partition(":")
        name, _, tun = tag.partition("@")              # "name@16" = this variant runs with the variant word --tuning | 16 (tests/lab_bits.py:
        tunings[tag] = lab_bits.split(a.tuning | int(tun or 0))      # a product tuning + experiment switches of include/ddcore_lab.h)
        libs.append((tag, build(name, [f for f in fl.split(",") if f]), []))
    ref = None
    for r in range(a.rounds + 1):
        for tag, lib, times in libs:
            builder.cursor.zero_()
            cb.tuning = tunings[tag][0]
            lib.dd_debug_tuning(tunings[tag][1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.dd_unproject_compact(C.byref(cb), C.byref(out), offs.data_ptr(), builder.cursor.data_ptr(), ws.data_ptr(), ws.numel(), stream)
            e1.record()
            lib.dd_debug_tuning(0)
            torch.cuda.synchronize()
            assert rc == 0, (tag, rc)
            if r == 0:
                chk = (int(builder.cursor.item()), float(builder.xyz[: int(builder.cursor.item())].double().sum()))
                ref = ref or (None if tag.startswith("x_") else chk)
                assert chk == ref or tag.startswith("x_"), f"variant {tag} computes something else: {chk} vs {ref}"      # (x_...: an experiment that is allowed to)
            else:
                times.append(e0.elapsed_time(e1))
    if a.stamps:
        tiles = (H * W + 4095) // 4096 * a.views          # (room for the finest tiling: the small-batch tile is 6144 pixels)
        st = torch.zeros((tiles, 16), dtype=torch.int64, device=dev)
        cb.refined_out = st.data_ptr()
        for tag, lib, _ in libs:
            if "stamp" not in tag:
                continue
            for rep in range(2):
                st.zero_(); builder.cursor.zero_()
                cb.tuning = tunings[tag][0]
                lib.dd_debug_tuning(tunings[tag][1])
                rc = lib.dd_unproject_compact(C.byref(cb), C.byref(out), offs.data_ptr(), builder.cursor.data_ptr(), ws.data_ptr(), ws.numel(), stream)
                lib.dd_debug_tuning(0)
                torch.cuda.synchronize()
            s_ = st.cpu().numpy().astype(np.float64)
            ok = s_[:, 4] > 0
            s_ = s_[ok]
            t0 = s_[:, 0]
            names = ["w0 loads+bits", "w0 barrier #1", "w0 look-back (incl. list barrier)", "w0 process", "w5 loads+bits", "w5 -> list done", "w5 wait look-back", "w5 process", "tile total (w0)"]
            seg = [s_[:, 1] - t0, s_[:, 2] - s_[:, 1], s_[:, 3] - s_[:, 2], s_[:, 4] - s_[:, 3],
                   s_[:, 9] - s_[:, 8], np.where(s_[:, 12] > 0, s_[:, 12] - s_[:, 9], 0), s_[:, 10] - np.where(s_[:, 12] > 0, s_[:, 12], s_[:, 9]), s_[:, 11] - s_[:, 10], s_[:, 4] - t0]
            names += ["w0 entry -> ticket known"]
            seg += [t0 - s_[:, 5]]
            rt0, rt1 = s_[:, 6], s_[:, 7]                       # 100 MHz constant clock, chip-wide
            span_us = (rt1.max() - rt0.min()) / 100.0
            life_us = (rt1 - rt0) / 100.0
            print(f"   workgroup lifetime (entry -> end of tile): median {np.median(life_us):.2f} us, mean {life_us.mean():.2f} us; kernel span {span_us:.1f} us; "
                  f"workgroups alive on average {life_us.sum() / span_us:.1f} of 512 slots; shader clock {np.median((s_[:, 4] - s_[:, 5]) / np.maximum(life_us, 1e-9)) / 1000:.2f} GHz")
            span = s_[:, 4].max() - t0.min()
            print(f"stamps {tag}: {len(s_)} tiles, kernel span {span:.0f} ticks; per tile, median / mean ticks:")
            for nm, x in zip(names, seg):
                print(f"   {nm:36s} {np.median(x):9.0f} {x.mean():9.0f}")
            print(f"   tiles in flight (sum of tile totals / span): {seg[-1].sum() / span:.1f}")
        cb.refined_out = None
    n = int(builder.cursor.item())
    alg = bench.algorithmic_bytes(cfg, a.views, n, False)
    print(f"{a.workload} x {a.views} views, {n} points, algorithmic {alg / 1e9:.3f} GB")
    base = float(np.median(libs[0][2]))
    for tag, _, times in libs:
        med = float(np.median(times))
        print(f"{tag:14s} median {med:8.3f} ms  min {min(times):8.3f}  frac {alg / med / 1e6 / 8000:6.3f}  vs first {base / med:6.3f}x")


if __name__ == "__main__":
    main()
