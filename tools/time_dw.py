"""developer aid: device time of the depthwise kernels per EfficientNet-B4 shape (batch 64), 10 launches per graph replay, through the C ABI.
usage: [SRBH_LIB_PATH=variant.so] python tools/time_dw.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srbh_amd import _lib
dev, B = "cuda:0", 64
L = _lib.lib()
shapes = [(48, 32, 3, 1), (144, 32, 3, 2), (192, 16, 3, 1), (192, 16, 5, 2), (336, 8, 5, 1), (336, 8, 3, 2), (672, 4, 3, 1), (672, 4, 5, 1), (960, 4, 5, 2), (1632, 2, 5, 1), (2688, 2, 3, 1)]


def timed(fn):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 100 * 1e3


for C, H, K, s in shapes:
    pl, pr, pt, pb = {(3, 1): (1, 1, 1, 1), (3, 2): (0, 1, 0, 1), (5, 1): (2, 2, 2, 2), (5, 2): (1, 2, 1, 2)}[(K, s)]
    OH = (H + pt + pb - K) // s + 1
    x = torch.randn(B, C, H, H, device=dev); w = torch.randn(C, 1, K, K, device=dev)
    y = torch.empty(B, C, OH, OH, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    ws = torch.empty(L.srbh_dwconv_bwd_weight_splits(B, C) * C * K * K, device=dev)
    geo = (B, C, H, H, K, s, pt, pl, OH, OH)
    f = timed(lambda: _lib.check(L.srbh_dwconv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), *geo, _lib.stream_ptr())))
    d = timed(lambda: _lib.check(L.srbh_dwconv_bwd_data(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), *geo, _lib.stream_ptr())))
    g = timed(lambda: _lib.check(L.srbh_dwconv_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), *geo, _lib.stream_ptr())))
    print(f"C={C:5d} {H:2d}x{H:<2d} K={K} s={s}: fwd {f:6.1f}  bwd_data {d:6.1f}  bwd_weight(+reduce) {g:6.1f} us", flush=True)
