"""developer probe: how long the GPU waits for the host at the step boundary (end of backward -> Adam -> the next step's first kernels).
Events are recorded at the host points; elapsed(a, b) between two events with nothing but one small kernel in between is the host's time.
python tools/step_boundary_probe.py [B]"""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd import synth
from srbh_amd.harness import TrainStep, synthetic_batch
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.rrdbnet import RRDBNet
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337))
torch.manual_seed(0)
net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
ts = TrainStep(net_hr.to(dev), net.to(dev), dev, status_every=0)
batch = synthetic_batch(B, 1, dev)
for _ in range(5): ts(batch, next_batch=batch)
opt = ts.optimizer
orig_step = opt.step
marks = {}


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


def step_wrapped(*a, **k):
    marks["pre_adam"] = (ev(), time.perf_counter())
    r = orig_step(*a, **k)
    marks["post_adam"] = (ev(), time.perf_counter())
    return r


opt.step = step_wrapped
rows = []
for i in range(12):
    marks.clear()
    t0 = time.perf_counter()
    e0 = ev()
    ts(batch, next_batch=batch)
    e1 = ev()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    rows.append((e0.elapsed_time(marks["pre_adam"][0]), marks["pre_adam"][0].elapsed_time(marks["post_adam"][0]), marks["post_adam"][0].elapsed_time(e1),
                 (marks["pre_adam"][1] - t0) * 1e3, (marks["post_adam"][1] - marks["pre_adam"][1]) * 1e3, (t1 - marks["post_adam"][1]) * 1e3))
rows = rows[2:]
n = len(rows)
m = [sum(r[i] for r in rows) / n for i in range(6)]
print(f"device ms: step start -> Adam issued {m[0]:.2f} | across optimizer.step() {m[1]:.2f} (the Adam kernel itself is ~0.1) | Adam -> step end {m[2]:.2f}")
print(f"host   ms: step start -> Adam issued {m[3]:.2f} | optimizer.step() {m[4]:.2f} | after it {m[5]:.2f}   (synchronised between steps: host never runs ahead of the previous step)")
