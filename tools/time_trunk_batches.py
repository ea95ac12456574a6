import ctypes, sys, time
sys.path.insert(0, '.')
import torch
from srbh_amd import _lib, synth
from srbh_amd.rrdbnet import RRDBNet
net = RRDBNet(3, 3); net.load_state_dict(synth.rrdbnet_state_dict(seed=1337, mode="init")); net = net.cuda().eval()
L = _lib.lib()
for B in (32, 64, 128, 32):
    x = synth.tiles(B, 8, 64, seed=1337)[:, :3].contiguous().cuda()
    with torch.no_grad():
        for _ in range(5): net.forward_feature(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): net.forward_feature(x)
        e1.record(); torch.cuda.synchronize()
        whole = e0.elapsed_time(e1) / 20
        L.srbh_trunk_timing(1); acc = []
        for _ in range(10):
            net.forward_feature(x); ms = ctypes.c_float(0.0); L.srbh_trunk_last_ms(ctypes.byref(ms)); acc.append(ms.value)
        L.srbh_trunk_timing(0)
    acc.sort()
    print("B=%d forward %.3f ms (%.3f per 32)  trunk %.3f ms (%.3f per 32)" % (B, whole, whole * 32 / B, acc[5], acc[5] * 32 / B))
