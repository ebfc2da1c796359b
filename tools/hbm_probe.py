#!/usr/bin/env python3
"""Practical HBM ceiling of the box: device-to-device copy / read-only / write-only rates (GB/s)."""
import torch

def rate(fn, nbytes, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9

def main():
    n = 1 << 30            # 1 Gi floats = 4 GiB per buffer
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    out = {}
    out["copy_rw_GBps"] = rate(lambda: b.copy_(a), 2 * a.numel() * 4)
    out["fill_w_GBps"] = rate(lambda: b.fill_(1.0), a.numel() * 4)
    out["sum_r_GBps"] = rate(lambda: a.sum(), a.numel() * 4)
    out["add_rrw_GBps"] = rate(lambda: torch.add(a, b, out=b), 3 * a.numel() * 4)
    print({k: round(v, 1) for k, v in out.items()})
    return out

if __name__ == "__main__":
    main()
