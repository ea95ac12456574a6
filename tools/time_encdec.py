"""developer timing: stock-op encoder + decoders fwd+bwd at B=64 under MIOpen search / memory-format choices"""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd.models import SRRegress_Cls_feature
dev = 'cuda:0'
B = 64
torch.manual_seed(0)
x = torch.rand(B, 8, 64, 64, device=dev)
def T(fn, n=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for bench in (False, True):
    for cl in (False, True):
        torch.backends.cudnn.benchmark = bench
        net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7).to(dev)
        xx = x
        if cl:
            net.encoder.to(memory_format=torch.channels_last); net.decoder1.to(memory_format=torch.channels_last); net.decoder2.to(memory_format=torch.channels_last)
            xx = x.contiguous(memory_format=torch.channels_last)
        def fb():
            f = net.encoder(xx); d = net.decoder1(*f); e = net.decoder2(*f); (d.sum() + e.sum()).backward()
        def fwd():
            with torch.no_grad():
                f = net.encoder(xx); net.decoder1(*f); net.decoder2(*f)
        print(f"cudnn.benchmark={bench} channels_last={cl}: enc+2dec fwd {T(fwd):.1f} ms | fwd+bwd {T(fb):.1f} ms", flush=True)
