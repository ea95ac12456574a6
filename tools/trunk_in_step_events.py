"""developer probe: the persistent trunk's launch time INSIDE the eager training step, by HIP events on its stream (libsrbh's hook, no profiler),
next to the same two launches back to back.  python tools/trunk_in_step_events.py   (SRBH_HRFEAT_FIRST=0|1 to compare issue orders)"""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from srbh_amd import _lib, synth
from srbh_amd.harness import TrainStep, synthetic_batch
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.rrdbnet import RRDBNet
dev = "cuda:0"
L = _lib.lib()
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337))
torch.manual_seed(0)
net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
ts = TrainStep(net_hr.to(dev), net.to(dev), dev, status_every=0)
batch = synthetic_batch(64, 1, dev)
for _ in range(5):
    ts(batch)
torch.cuda.synchronize()
L.srbh_trunk_timing(1)
acc = []
ts(batch)
for _ in range(12):
    ts(batch)                      # the whole next step is queued before the previous trunk time is read
    ms = ctypes.c_float(0.0)
    _lib.check(L.srbh_trunk_last_ms(ctypes.byref(ms)))
    acc.append(ms.value)
torch.cuda.synchronize()
x = batch[0][:, :3].contiguous()
iso = []
with torch.no_grad():
    for _ in range(10):
        net_hr.forward_feature(x, out_dtype=torch.float16)
        ms = ctypes.c_float(0.0)
        _lib.check(L.srbh_trunk_last_ms(ctypes.byref(ms)))
        iso.append(ms.value)
L.srbh_trunk_timing(0)
acc.sort(); iso.sort()
print("trunk, two launches (B=64), ms: inside the eager training step median %.3f (min %.3f max %.3f) | back to back median %.3f" % (acc[len(acc) // 2], acc[0], acc[-1], iso[len(iso) // 2]))
