"""developer probe: do chains of tiny dependent kernels run slower once the power management has clocked the chip down?  A chain of 1400 small
elementwise kernels, timed in segments of 100 (HIP events), (a) right behind two persistent-trunk launches (8 ms of matrix-core work: clocks up),
(b) behind 20 ms of the same tiny kernels (clocks down).  Also the real thing: the encoder forward+backward of the height model behind either.
python tools/tiny_clock_probe.py"""
import sys, time
sys.path.insert(0, '.')
import torch
from srbh_amd import synth
from srbh_amd.rrdbnet import RRDBNet
from srbh_amd.models import SRRegress_Cls_feature
dev = "cuda:0"
net = RRDBNet(3, 3); net.load_state_dict(synth.rrdbnet_state_dict(seed=1337, mode="init")); net = net.cuda().eval()
x = synth.tiles(64, 8, 64, seed=1337)[:, :3].contiguous().cuda()
small = torch.rand(4096, device=dev)


def tiny(n):
    t = small
    for _ in range(n):
        t = t * 1.0001 + 0.5
    return t


def segments(pre):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(15)]
    torch.cuda.synchronize()
    with torch.no_grad():
        if pre == "trunk":
            net.forward_feature(x)
        else:
            tiny(3000)
        ev[0].record()
        for k in range(14):
            tiny(100)
            ev[k + 1].record()
    torch.cuda.synchronize()
    return [ev[k].elapsed_time(ev[k + 1]) * 10 for k in range(14)]      # us per kernel


with torch.no_grad():
    for _ in range(3):
        net.forward_feature(x); tiny(500)
for pre in ("trunk", "tiny", "trunk", "tiny"):
    s = segments(pre)
    print(f"behind {pre:5s}: us per tiny kernel in 14 segments of 100: " + " ".join(f"{v:.2f}" for v in s))

# the encoder + decoders of the height model, forward + backward (training mode), behind either
torch.manual_seed(0)
model = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7).to(dev).train()
lr = torch.rand(64, 8, 64, 64, device=dev)


def encdec():
    f = model.encoder(lr)
    d1 = model.decoder1(*f)
    d2 = model.decoder2(*f)
    (d1.sum() + d2.sum()).backward()


from srbh_amd import hrfuse as H
with H.head_precision("f16"):
    for _ in range(3):
        encdec()
    for pre in ("trunk", "tiny", "trunk", "tiny"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.no_grad():
            if pre == "trunk":
                net.forward_feature(x)
            else:
                tiny(3000)
        e0.record()
        encdec()
        e1.record()
        torch.cuda.synchronize()
        print(f"encoder + decoders fwd+bwd (B=64) behind {pre:5s}: {e0.elapsed_time(e1):.2f} ms")
