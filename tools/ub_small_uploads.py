import time, torch, numpy as np
dev = torch.device("cuda", 0)
big_h = torch.empty(41 << 20, dtype=torch.uint8, pin_memory=True)
small_p = torch.empty(1 << 12, dtype=torch.uint8, pin_memory=True)
small_np = np.zeros(1 << 12, np.uint8)
x = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
def t(fn, n=50):
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        big_d = big_h.to(dev, non_blocking=True)       # 41 MB DMA in flight
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    return np.median(ts) * 1e6
print("pinned small .to non_blocking behind a 41 MB DMA: %.0f us" % t(lambda: small_p.to(dev, non_blocking=True)))
print("pageable small .to behind a 41 MB DMA:            %.0f us" % t(lambda: torch.from_numpy(small_np).to(dev)))
print("pageable small .to non_blocking:                  %.0f us" % t(lambda: torch.from_numpy(small_np).to(dev, non_blocking=True)))
d = torch.empty(1 << 12, dtype=torch.uint8, device=dev)
print("copy_ into existing device tensor from pinned:    %.0f us" % t(lambda: d.copy_(small_p, non_blocking=True)))
print("kernel launch (fill) behind DMA:                  %.0f us" % t(lambda: d.fill_(1)))
ev = torch.cuda.Event()
print("event record:                                     %.0f us" % t(lambda: ev.record()))
torch.cuda.synchronize()
t0=time.perf_counter(); 
for _ in range(100): small_p.to(dev, non_blocking=True)
print("pinned small .to idle stream: %.1f us each" % ((time.perf_counter()-t0)/100*1e6))
