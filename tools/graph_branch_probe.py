"""developer aid: does a HIP graph captured from two forked streams run its branches concurrently on this ROCm?  Two chains of small,
latency-bound kernels (a few workgroups each) captured (a) on one stream, (b) forked over two streams; replay time of each.
python tools/graph_branch_probe.py (GPU)"""
import torch
dev = torch.device("cuda:0")
a = torch.rand(64 * 1024, device=dev)
b = torch.rand(64 * 1024, device=dev)
N = 200


def chain(t):
    for _ in range(N):
        t = t * 1.0001 + 0.5
    return t


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(2):
    chain(a); chain(b)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    chain(a); chain(b)
side = torch.cuda.Stream()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        yb = chain(b)
    ya = chain(a)
    cur.wait_stream(side)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    chain(a)
print("one chain            %.3f ms" % timeit(g3.replay))
print("two chains, 1 stream %.3f ms" % timeit(g1.replay))
print("two chains, forked   %.3f ms" % timeit(g2.replay))


def eager_forked():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain(b)
    chain(a)
    cur.wait_stream(side)


print("eager, forked        %.3f ms" % timeit(eager_forked, 5))
print("eager, 1 stream      %.3f ms" % timeit(lambda: (chain(a), chain(b)), 5))

g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4):
    chain(b)


def two_graphs():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        g4.replay()
    g3.replay()
    cur.wait_stream(side)


print("two graphs, two streams %.3f ms" % timeit(two_graphs))
big = torch.rand(256 * 1024 * 1024 // 4, device=dev)


def big_chain():
    t = big
    for _ in range(10):
        t = t * 1.0001
    return t


big_chain(); torch.cuda.synchronize()
g5 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g5):
    big_chain()


def big_plus_small():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        g4.replay()
    g5.replay()
    cur.wait_stream(side)


print("HBM-bound chain alone (10 x 256 MB r+w) %.3f ms" % timeit(g5.replay))
print("HBM-bound chain + small chain on a second stream %.3f ms" % timeit(big_plus_small))
