"""developer probe: where the time of ONE steady pipelined training step goes, UNPROFILED (rocprofv3 inflates the ~750 small kernels):
HIP events recorded on the main stream at the phase boundaries of the step -- in front of / behind each sub-network's forward, when the
gradient of each sub-network's output arrives (= its backward is about to start), behind backward and behind Adam -- averaged over steps.
python tools/phase_events.py [B] [steps]"""
import sys, time, collections, torch
sys.path.insert(0, '.')
from srbh_amd import synth
from srbh_amd.harness import TrainStep, synthetic_batch
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.rrdbnet import RRDBNet
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337))
torch.manual_seed(0)
net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
ts = TrainStep(net_hr.to(dev), net.to(dev), dev, status_every=0)
batch = synthetic_batch(B, 1, dev)
for _ in range(5): ts(batch, next_batch=batch)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))


def wrap(mod, name):
    orig = mod.forward

    def fwd(*a, **k):
        mark("fwd " + name + " >")
        out = orig(*a, **k)
        mark("fwd " + name + " <")
        o = out[-1] if isinstance(out, (list, tuple)) else out
        if torch.is_tensor(o) and o.requires_grad:
            o.register_hook(lambda g, n=name: mark("bwd " + n + " >"))
        return out
    mod.forward = fwd


for n in ("encoder", "decoder1", "decoder2", "hrfeat", "reg", "seg"):
    wrap(getattr(net, n), n)
opt_step = ts.optimizer.step


def step_wrapped(*a, **k):
    mark("adam >"); r = opt_step(*a, **k); mark("adam <"); return r


ts.optimizer.step = step_wrapped
acc = collections.OrderedDict()
torch.cuda.synchronize(); t0 = time.perf_counter()
allmarks = []
for _ in range(steps):
    marks.clear()
    mark("step >")
    ts(batch, next_batch=batch)
    mark("step <")
    allmarks.append(list(marks))
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
for ms in allmarks[2:]:
    for (n0, e0), (n1, e1) in zip(ms[:-1], ms[1:]):
        k = n0 + "  ->  " + n1
        acc[k] = acc.get(k, 0.0) + e0.elapsed_time(e1)
n = len(allmarks) - 2
print(f"B={B}: {wall:.2f} ms per step (wall, {steps} steps); main-stream intervals, ms:")
tot = 0.0
for k, v in acc.items():
    print(f"  {v / n:7.3f}  {k}")
    tot += v / n
print(f"  {tot:7.3f}  sum")
