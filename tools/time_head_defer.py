"""developer timing: the HR head's forward + backward (HRfeature + the two HRfuse_residual heads, f16 training mode, batch 64) with the
weight-gradient reduces deferred to one pair of launches per block chain (SRBH_WGRAD_DEFER) and without -- interleaved in ONE process,
device time by events (the training-step bench is too noisy on a shared box to resolve 0.3 ms)."""
import sys, torch
sys.path.insert(0, '.')
from srbh_amd import hrfuse as H, hrfuse_autograd as HA
from srbh_amd.hrfuse import HRfeature, HRfuse_residual
dev = 'cuda:0'; B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
hf = HRfeature(64, 16, 16).to(dev).train()
reg = HRfuse_residual(16, 16, 16, 1, 4).to(dev).train()
seg = HRfuse_residual(16, 16, 16, 7, 4).to(dev).train()
fea = torch.randn(B, 64, 256, 256, device=dev).contiguous(memory_format=torch.channels_last).half()
lo = torch.randn(B, 16, 64, 64, device=dev, requires_grad=True)


def step():
    with H.head_precision("f16"):
        s = hf(fea); h = reg(lo, s); b = seg(lo, s)
        (h.sum() + b.sum()).backward()


def T(n=10):
    for _ in range(3): step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): step()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


for rnd in range(4):
    r = []
    for d in (False, True):
        HA.WGRAD_DEFER = d
        r.append(T())
    print(f"round {rnd}: head fwd+bwd  reduces at once {r[0]:.3f} ms | deferred {r[1]:.3f} ms")
