import os, sys, time, torch
sys.path.insert(0, '.')
os.environ["SRBH_PT_PROF"] = "1"
from srbh_amd import synth
from srbh_amd.rrdbnet import RRDBNet
sd = synth.rrdbnet_state_dict(seed=1337, mode="init")
net = RRDBNet(3, 3); net.load_state_dict(sd); net = net.cuda().eval()
for kind in ("real", "zeros"):
    x = synth.tiles(32, 8, 64, seed=1337)[:, :3].contiguous().cuda()
    if kind == "zeros": x.zero_()
    with torch.no_grad():
        for _ in range(3): net.forward_feature(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): net.forward_feature(x)
        torch.cuda.synchronize(); print(kind, "ms/forward (incl. prof sync)", (time.perf_counter() - t0) / 5 * 1e3)
