#!/usr/bin/env python3
"""points3D.bin of a dense cloud: host writer (device->host copy, float64 arrays, record array, tofile) vs the
streaming writer (records formatted on the GPU, pinned double buffers, copies overlapped with the file writes).  GPU box.

    python tools/bench_model_write.py [--points 64] [--big-points 326] [--dir /tmp/dd_model_write]
"""
import argparse, os, shutil, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import depthdensifier_amd as dd
from depthdensifier_amd.colmap_io import Reconstruction

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=float, default=64.0, help="millions of dense points for the host-vs-streaming comparison")
ap.add_argument("--big-points", type=float, default=326.0, help="millions of points for the streaming-only run (garden at density 1); 0 = skip")
ap.add_argument("--dir", type=Path, default=Path("/tmp/dd_model_write"))
a = ap.parse_args()
dev = torch.device("cuda", 0)
a.dir.mkdir(parents=True, exist_ok=True)


def cloud_of(n):
    g = torch.Generator(device=dev).manual_seed(1)
    rec = torch.empty((n, 4), dtype=torch.float32, device=dev)
    rec[:, :3].normal_(generator=g)
    rec.view(torch.int32)[:, 3] = torch.randint(0, 1 << 24, (n,), generator=g, device=dev, dtype=torch.int32) | (0xFF << 24)
    return dd.FusedCloud.from_packed(rec, torch.tensor([0, n], dtype=torch.int64, device=dev))


def timed(label, fn, nbytes):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:58s} {dt:7.2f} s  {nbytes / dt / 1e9:6.2f} GB/s of file", flush=True)
    return dt


n = int(a.points * 1e6)
cloud = cloud_of(n)
size = n * 51
print(f"{n / 1e6:.0f} M points -> {size / 1e9:.2f} GB of records; directory {a.dir} ({shutil.disk_usage(a.dir).free / 1e9:.0f} GB free)")


def host():
    rec = Reconstruction()
    rec.add_points3D(cloud.points.cpu().numpy().astype(np.float64), cloud.colors.cpu().numpy())
    rec.write_binary(a.dir / "host")


def stream():
    Reconstruction().write_binary(a.dir / "stream", dense=cloud)


for rep in range(2):
    timed("host writer (round 1: D2H, float64, record array, tofile)", host, size)
    timed("streaming writer (GPU-formatted records, overlapped)", stream, size)
same = (a.dir / "host" / "points3D.bin").read_bytes() == (a.dir / "stream" / "points3D.bin").read_bytes()
print("files identical:", same)
assert same
shutil.rmtree(a.dir / "host"); shutil.rmtree(a.dir / "stream")
del cloud
torch.cuda.empty_cache()
if a.big_points > 0:
    n = int(a.big_points * 1e6)
    if shutil.disk_usage(a.dir).free < n * 51 * 1.2:
        print(f"skipping the {n / 1e6:.0f} M-point run: not enough room in {a.dir}")
    else:
        cloud = cloud_of(n)
        timed(f"streaming writer, {n / 1e6:.0f} M points ({n * 51 / 1e9:.1f} GB)", lambda: Reconstruction().write_binary(a.dir / "big", dense=cloud), n * 51)
        shutil.rmtree(a.dir / "big")
shutil.rmtree(a.dir, ignore_errors=True)
