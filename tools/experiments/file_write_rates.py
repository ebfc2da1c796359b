#!/usr/bin/env python3
"""How fast do the bytes of points3D.bin reach a fresh file on this box?  One thread writing sequentially (what model_writer does), and
1 / 2 / 4 / 8 threads with pwrite on disjoint stretches of chunks; 16 GiB from a 204 MiB buffer each.   python tools/experiments/file_write_rates.py [dir]"""
import os, sys, tempfile, threading, time
import numpy as np
d = sys.argv[1] if len(sys.argv) > 1 else tempfile.gettempdir()
chunk = np.random.default_rng(0).integers(0, 256, 204 << 20, dtype=np.uint8)
mv = memoryview(chunk)
total_chunks = 80                                     # 16 GiB
def run(threads, how):
    path = os.path.join(d, f"dd_write_probe_{os.getpid()}.bin")
    if os.path.exists(path): os.unlink(path)
    t0 = time.perf_counter()
    if how == "write":
        with open(path, "wb") as f:
            for _ in range(total_chunks): f.write(mv)
    else:
        fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
        def work(k):
            for c in range(k, total_chunks, threads):
                off, done = c * len(mv), 0
                while done < len(mv): done += os.pwrite(fd, mv[done:], off + done)
        ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        for t in ts: t.start()
        for t in ts: t.join()
        os.close(fd)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter(); os.unlink(path); t_rm = time.perf_counter() - t1
    print(f"{how:6s} {threads} thread(s): {total_chunks * len(mv) / dt / 1e9:6.2f} GB/s ({dt:5.2f} s for {total_chunks * len(mv) / 2**30:.0f} GiB); unlink {t_rm:.2f} s", flush=True)
print("directory", d, "free GiB", os.statvfs(d).f_bavail * os.statvfs(d).f_frsize / 2**30)
run(1, "write")
for n in (1, 2, 4, 8, 16): run(n, "pwrite")
run(1, "write")
