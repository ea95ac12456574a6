"""developer aid: per-launch floor of this box -- eager torch op, the same ops replayed from one HIP graph, a MIOpen conv."""
import time, torch
import torch.nn.functional as F
d = torch.device("cuda:0")
x = torch.zeros(1024, device=d)
def t(fn, n=2000):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    ti = time.perf_counter() - t0
    torch.cuda.synchronize(); return 1e6 * ti / n, 1e6 * (time.perf_counter() - t0) / n
print("eager add_: issue %.2f us, wall %.2f us per launch" % t(lambda: x.add_(1)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): x.add_(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(1000): x.add_(1)
torch.cuda.synchronize()
a, b = t(g.replay, 20)
print("graph of 1000 add_: wall %.2f us per node" % (b / 1000))
xc = torch.randn(64, 96, 16, 16, device=d, requires_grad=True); w = torch.randn(24, 96, 1, 1, device=d, requires_grad=True)
print("conv1x1 fwd (MIOpen): issue %.2f us, wall %.2f us" % t(lambda: F.conv2d(xc, w), 500))
def fb():
    y = F.conv2d(xc, w); y.backward(y)
print("conv1x1 fwd+bwd (MIOpen): issue %.2f us, wall %.2f us" % t(fb, 300))
w2 = w.detach().view(24, 96).requires_grad_()
def fb2():
    y = torch.matmul(w2, xc.flatten(2)); y.backward(y)
print("matmul 1x1 fwd+bwd: issue %.2f us, wall %.2f us" % t(fb2, 300))
bn = torch.nn.BatchNorm2d(96).to(d)
def fb3():
    y = F.silu(bn(xc)); y.backward(y)
print("bn+silu fwd+bwd: issue %.2f us, wall %.2f us" % t(fb3, 300))
