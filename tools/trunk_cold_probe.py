import ctypes, sys
sys.path.insert(0, '.')
import torch
from srbh_amd import _lib, synth
from srbh_amd.rrdbnet import RRDBNet
net = RRDBNet(3, 3); net.load_state_dict(synth.rrdbnet_state_dict(seed=1337, mode="init")); net = net.cuda().eval()
L = _lib.lib()
B = 64
x = synth.tiles(B, 8, 64, seed=1337)[:, :3].contiguous().cuda()
big = torch.empty(512 << 20, dtype=torch.float32, device="cuda")      # 2 GB
big2 = torch.empty_like(big)
def run(thrash, n=10):
    acc = []
    with torch.no_grad():
        L.srbh_trunk_timing(1)
        for _ in range(n):
            if thrash == "copy": big2.copy_(big)
            if thrash == "copy3":
                for _ in range(3): big2.copy_(big)
            net.forward_feature(x); ms = ctypes.c_float(0.0); L.srbh_trunk_last_ms(ctypes.byref(ms)); acc.append(ms.value)
        L.srbh_trunk_timing(0)
    acc.sort(); return acc[len(acc) // 2]
with torch.no_grad():
    for _ in range(5): net.forward_feature(x)
print("trunk B=64 ms: isolated %.3f | after a 2 GB copy %.3f | after three (4.5 ms of HBM streaming) %.3f | isolated again %.3f" % (run(None), run("copy"), run("copy3"), run(None)))
