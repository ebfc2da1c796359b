#!/usr/bin/env python3
"""Sharded model write: N processes (stand-ins for the ranks of a multi-GPU scan; here they share the box's GPU) each
stream their slice of the dense records into ONE points3D.bin in place (model_writer.write_dense_at).  GPU box.

    python tools/bench_model_write_sharded.py [--points 256] [--ranks 1 2 4]
"""
import argparse, os, shutil, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=float, default=256.0, help="millions of dense points in all")
ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4])
ap.add_argument("--dir", type=Path, default=Path("/tmp/dd_model_write_sharded"))
ap.add_argument("--worker", nargs=4, default=None, help=argparse.SUPPRESS)      # path, offset, first id, points
a = ap.parse_args()

if a.worker:
    import torch
    import depthdensifier_amd as dd
    from depthdensifier_amd.model_writer import write_dense_at
    path, off, first, n = a.worker[0], int(a.worker[1]), int(a.worker[2]), int(a.worker[3])
    dev = torch.device("cuda", 0)
    rec = torch.empty((n, 4), dtype=torch.float32, device=dev)
    rec[:, :3].normal_()
    rec.view(torch.int32)[:, 3] = (0x00C86432 | (0xFF << 24)) - (1 << 32)
    cloud = dd.FusedCloud.from_packed(rec, torch.tensor([0, n], dtype=torch.int64, device=dev))
    torch.cuda.synchronize()
    print("READY", flush=True)
    sys.stdin.readline()                                   # all workers start writing together
    t0 = time.perf_counter()
    write_dense_at(path, off, cloud, first)
    torch.cuda.synchronize()
    print(f"DONE {time.perf_counter() - t0:.3f}", flush=True)
    sys.exit(0)

from depthdensifier_amd.colmap_io import Reconstruction
n = int(a.points * 1e6)
for R in a.ranks:
    out = a.dir / f"r{R}"
    shutil.rmtree(out, ignore_errors=True)
    where = Reconstruction().write_binary(out, dense=None, dense_total=n)
    cuts = [n * r // R for r in range(R + 1)]
    procs = [subprocess.Popen([sys.executable, __file__, "--worker", str(out / "points3D.bin"), str(where["dense_offset"] + cuts[r] * 51),
                               str(where["first_dense_id"] + cuts[r]), str(cuts[r + 1] - cuts[r])],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for r in range(R)]
    for p in procs:
        assert p.stdout.readline().strip() == "READY"
    t0 = time.perf_counter()
    for p in procs:
        p.stdin.write("go\n"); p.stdin.flush()
    per = [float(p.stdout.readline().split()[1]) for p in procs]
    wall = time.perf_counter() - t0
    for p in procs:
        p.wait()
    size = (out / "points3D.bin").stat().st_size
    print(f"{R} writer(s): {n / 1e6:.0f} M points, {size / 1e9:.2f} GB in {wall:.2f} s = {size / wall / 1e9:.2f} GB/s   (slowest writer {max(per):.2f} s)", flush=True)
    shutil.rmtree(out, ignore_errors=True)
shutil.rmtree(a.dir, ignore_errors=True)
