"""developer timing: parts of the eval model at the city-inference batch size (B=128)"""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd import synth
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.rrdbnet import RRDBNet
from srbh_amd.mosaic import Mosaic
dev = 'cuda:0'; B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.backends.cudnn.benchmark = True
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337)); net_hr = net_hr.to(dev).eval()
torch.manual_seed(0)
model = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=False, chans_build=7).to(dev).eval()
x = torch.randn(B, 8, 64, 64, device=dev) * 0.25 + 0.35
def T(fn, k=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3, r
with torch.no_grad():
    t_r, fea = T(lambda: net_hr.forward_feature(x[:, :3]))
    t_m, out = T(lambda: model(x, fea))
    t_e, feats = T(lambda: model.encoder(x))
    t_h, sup = T(lambda: model.hrfeat(fea))
    t_d, hf = T(lambda: model.decoder1(*feats))
    t_g, _ = T(lambda: model.reg(hf, sup))
    t_s, _ = T(lambda: model.seg(hf, sup))
    m = Mosaic(4096, 4096, 7, dev)
    pos = [[(i % 16) * 48, (i // 16) * 48, 64, 64] for i in range(B)]
    t_a, _ = T(lambda: m.add(out[0], out[1], pos))
print(f"B={B}: rrdb {t_r:.2f} | model {t_m:.2f} = encoder {t_e:.2f} + hrfeat {t_h:.2f} + 2 x decoder {t_d:.2f} + reg {t_g:.2f} + seg {t_s:.2f} | mosaic add {t_a:.2f} ms")
