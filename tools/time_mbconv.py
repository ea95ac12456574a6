"""developer aid: per-shape device time of the training BatchNorm / squeeze-excite kernels (csrc/srbh_mbconv.hip) at the shapes of the
EfficientNet-B4 encoder on 64x64 tiles, batch 64, next to the stock ops.   usage: python tools/time_mbconv.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from torch import nn
from srbh_amd import mbconv_autograd as MB

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C, H, SQ in ((144, 16, 6), (192, 16, 8), (192, 8, 8), (336, 8, 14), (336, 4, 14), (672, 4, 28), (960, 4, 40), (960, 2, 40), (1632, 2, 68), (2688, 2, 112),
                 (56, 8, 0), (112, 4, 0), (272, 2, 0)):
    bn = nn.BatchNorm2d(C, momentum=0.01, eps=1e-3).to(dev).train()
    x = torch.randn(B, C, H, H, device=dev, requires_grad=True)
    gy = torch.randn(B, C, H, H, device=dev)

    def graphed(make):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            make()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            for _ in range(10):
                make()
        return lambda: g.replay()

    def ours():
        y = MB.bn_act_train(bn, x, "silu")
        y.backward(gy)

    def stock():
        y = F.silu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps))
        y.backward(gy)
    line = f"C={C:5d} {H}x{H}  bn+silu fwd+bwd: ours {timeit(graphed(ours)) / 10:7.1f} us  stock {timeit(graphed(stock)) / 10:7.1f} us"
    if SQ:
        red, exp = nn.Conv2d(C, SQ, 1).to(dev), nn.Conv2d(SQ, C, 1).to(dev)

        def ours_se():
            y = MB.bn_swish_se_train(bn, x, red, exp)
            y.backward(gy)

        def stock_se():
            s_ = F.silu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps))
            y = torch.sigmoid(exp(F.silu(red(F.adaptive_avg_pool2d(s_, 1))))) * s_
            y.backward(gy)
        line += f" | bn+silu+SE fwd+bwd: ours {timeit(graphed(ours_se)) / 10:7.1f} us  stock {timeit(graphed(stock_se)) / 10:7.1f} us"
    print(line, flush=True)
