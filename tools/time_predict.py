"""developer timing of the sharded predict path (BASELINE config 5 shape): RRDBNet features -> eval head -> device mosaic."""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd import synth
from srbh_amd.harness import predict_tiles
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.mosaic import Mosaic
from srbh_amd.rrdbnet import RRDBNet
dev = 'cuda:0'
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337)); net_hr = net_hr.to(dev).eval()
torch.manual_seed(0)
model = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=False, chans_build=7).to(dev).eval()
n = 256
tiles = synth.tiles(n, 8, 64, seed=3, kind="grid")
gw = 16
pos = [[(i % gw) * 48, (i // gw) * 48, 64, 64] for i in range(n)]
H, W = ((n // gw) * 48 + 16) * 4, (gw * 48 + 16) * 4
tiles_d = tiles.to(dev)
def city():
    m = Mosaic(H, W, 7, dev)
    predict_tiles(net_hr, model, tiles_d, pos, m, batch=32)
    return m.finalize()
city(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); city(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = sorted(ts)[1]
print(f"predict: {n} tiles in {t*1e3:.1f} ms -> {n/t:.0f} tiles/s (RRDB alone would be ~{n/5600*1e3:.0f} ms)")
with torch.no_grad():
    x = tiles_d[:32]
    def T(fn, k=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3, r
    t_r, fea = T(lambda: net_hr.forward_feature(x[:, :3]))
    t_m, out = T(lambda: model(x, fea))
    m = Mosaic(H, W, 7, dev)
    t_a, _ = T(lambda: m.add(out[0], out[1], pos[:32]))
    print(f"per batch of 32: rrdb {t_r:.2f} ms | model eval {t_m:.2f} ms | mosaic add {t_a:.2f} ms")
with torch.no_grad():
    t_e, feats = T(lambda: model.encoder(x))
    t_h, sup = T(lambda: model.hrfeat(fea))
    t_d, hf = T(lambda: model.decoder1(*feats))
    t_g, _ = T(lambda: model.reg(hf, sup))
    t_s, _ = T(lambda: model.seg(hf, sup))
    print(f"  encoder {t_e:.2f} | hrfeat {t_h:.2f} | decoder {t_d:.2f} | reg {t_g:.2f} | seg {t_s:.2f} ms")
# HIP-graph capture of the eval model forward
with torch.no_grad():
    xs, fs = x.clone(), fea.clone()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): model(xs, fs)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = model(xs, fs)
    ref = model(xs, fs)
    g.replay(); torch.cuda.synchronize()
    print("graph == eager:", all(torch.equal(a, b) for a, b in zip(outs, ref)))
    t_g, _ = T(lambda: g.replay())
    print(f"  model eval eager {t_m:.2f} ms | graph replay {t_g:.2f} ms")
with torch.no_grad():
    t_e0, f0 = T(lambda: model.encoder(x))
    with torch.autocast("cuda", dtype=torch.float16):
        t_e16, f16 = T(lambda: model.encoder(x))
    enc_cl = model.encoder.to(memory_format=torch.channels_last)
    xcl = x.contiguous(memory_format=torch.channels_last)
    t_ecl, fcl = T(lambda: enc_cl(xcl))
    with torch.autocast("cuda", dtype=torch.float16):
        t_ecl16, _ = T(lambda: enc_cl(xcl))
    print(f"  encoder eval: fp32 NCHW {t_e0:.2f} | fp16 autocast {t_e16:.2f} | fp32 channels_last {t_ecl:.2f} | fp16 channels_last {t_ecl16:.2f} ms")
    print("  fp16 rel err of deepest feature:", float((f16[-1].float() - f0[-1]).norm() / f0[-1].norm()))
