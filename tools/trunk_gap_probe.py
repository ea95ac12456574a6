"""developer probe: what precedes the trunk launch and what it costs.  The persistent trunk kernel timed (HIP events on its stream, libsrbh's hook)
after (a) nothing -- back-to-back forwards, (b) ~5 ms of tiny dependent kernels (the encoder-backward phase of the training step: chip nearly idle),
(c) a host-side idle gap of 5 ms, (d) ~5 ms of HBM streaming (the head phase).  python tools/trunk_gap_probe.py"""
import ctypes, sys, time
sys.path.insert(0, '.')
import torch
from srbh_amd import _lib, synth
from srbh_amd.rrdbnet import RRDBNet
net = RRDBNet(3, 3); net.load_state_dict(synth.rrdbnet_state_dict(seed=1337, mode="init")); net = net.cuda().eval()
L = _lib.lib()
B = 64
x = synth.tiles(B, 8, 64, seed=1337)[:, :3].contiguous().cuda()
small = torch.rand(4096, device="cuda")
big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")      # 1 GB
big2 = torch.empty_like(big)


def run(mode, n=12):
    acc = []
    with torch.no_grad():
        L.srbh_trunk_timing(1)
        for _ in range(n):
            if mode == "tiny":
                t = small
                for _ in range(700):
                    t = t * 1.0001 + 0.5
            elif mode == "idle":
                torch.cuda.synchronize(); time.sleep(0.005)
            elif mode == "hbm":
                for _ in range(4):
                    big2.copy_(big)
            net.forward_feature(x); ms = ctypes.c_float(0.0); L.srbh_trunk_last_ms(ctypes.byref(ms)); acc.append(ms.value)
        L.srbh_trunk_timing(0)
    acc.sort(); return acc[len(acc) // 2]


with torch.no_grad():
    for _ in range(5): net.forward_feature(x)
print("trunk, B=64 (two launches), median ms: back to back %.3f | after ~5 ms of tiny kernels %.3f | after a 5 ms idle gap %.3f | after ~5 ms of HBM streaming %.3f | back to back again %.3f"
      % (run(None), run("tiny"), run("idle"), run("hbm"), run(None)))
