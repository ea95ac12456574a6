"""developer aid: SR-stage generator forward + backward (rrdbnet_autograd 'fast' mode, B tiles) as eager launches vs one HIP graph replay.
python tools/sr_graph_probe.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srbh_amd import rrdbnet_autograd as RA, synth
from srbh_amd.rrdbnet import RRDBNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
RA.set_train_precision("fast")
net = RRDBNet(3, 3, num_block=23)
net.load_state_dict(synth.rrdbnet_state_dict(num_block=23, seed=1337, mode="init"))
net = net.to(dev).train().enable_training_path(True)
x = synth.tiles(B, 8, 64, seed=1337)[:, :3].contiguous().to(dev)
w = None


def step():
    global w
    for p in net.parameters():
        p.grad = None
    y = net(x)
    if w is None:
        w = torch.randn(y.shape, generator=torch.Generator(device=dev).manual_seed(4242), device=dev)
    (y * w).sum().backward()


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    step()
print("eager  %.2f ms" % timeit(step))
g0 = float(sum(p.grad.double().pow(2).sum() for p in net.parameters() if p.grad is not None).sqrt())
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("graph  %.2f ms" % timeit(g.replay))
g1 = float(sum(p.grad.double().pow(2).sum() for p in net.parameters() if p.grad is not None).sqrt())
print("grad norm eager %.6g  graph %.6g" % (g0, g1))
