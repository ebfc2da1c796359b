import gc, sys, torch
sys.path.insert(0, "/root/repo")
import depthdensifier_amd as dd
from depthdensifier_amd import placement as pl
dev = torch.device("cuda", 0)
gb = lambda: round(torch.cuda.mem_get_info(dev)[0] / 2**30, 1)
print("free at start", gb())
b = dd.CloudBuilder(1 << 30, normals=True, colors=True, pixel_index=False, device=dev, placement="probed")
print("placed:", b.placement.mode, "free", gb())
del b
gc.collect(); torch.cuda.empty_cache()
print("after del + gc", gb(), pl._arenas[0].stats() if pl._arenas else None)
pl.trim(dev)
print("after trim", gb())
