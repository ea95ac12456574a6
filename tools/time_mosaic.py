"""developer aid: the device mosaic's cost per batch of 256 tiles and per city as a function of the city's size (accumulators of
(7 + 2) x H x W uint32: 2.7 GB for a 2 000-tile city, 27 GB for a 20 000-tile one).  usage: time_mosaic.py [batch=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from srbh_amd.mosaic import Mosaic
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda:0"
h = torch.rand((batch, 1, 256, 256), device=dev) * 30
b = torch.randn((batch, 7, 256, 256), device=dev).contiguous(memory_format=torch.channels_last)
for n in (500, 2000, 8000, 20000):
    gw = int(np.ceil(np.sqrt(n))); gh = (n + gw - 1) // gw
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = Mosaic((gh * 48 + 16) * 4, (gw * 48 + 16) * 4, 7, dev)
    torch.cuda.synchronize(); t_alloc = time.perf_counter() - t0
    nb = min(20, n // batch) or 1
    starts = np.linspace(0, max(0, n - batch), nb).astype(int)
    poss = [[[(i % gw) * 48, (i // gw) * 48, 64, 64] for i in range(s, s + batch)] for s in starts]
    m.add(h, b, poss[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in poss:
        m.add(h, b, p)
    torch.cuda.synchronize(); t_add = (time.perf_counter() - t0) / len(poss)
    t0 = time.perf_counter(); out = m.finalize(); torch.cuda.synchronize(); t_fin = time.perf_counter() - t0
    print("city of %5d tiles: mosaic %5.1f GB  alloc+zero %6.1f ms  add %.3f ms per batch of %d  finalize %6.1f ms  -> per 256 tiles: %.3f ms"
          % (n, (m.C + 2) * m.H * m.W * 4 / 1e9, t_alloc * 1e3, t_add * 1e3, batch, t_fin * 1e3, t_add * 1e3 + (t_alloc + t_fin) * 1e3 / (n / batch)), flush=True)
    del m, out
