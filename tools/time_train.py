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
ts = TrainStep(net_hr.to(dev), net.to(dev), dev)
batch = synthetic_batch(B, 1, dev)
for _ in range(3): ts(batch)
torch.cuda.synchronize()
def T(fn, n=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
lr = batch[0]
with torch.no_grad():
    t_rrdb, fea = T(lambda: net_hr.forward_feature(lr[:, :3]))
    t_enc, feats = T(lambda: net.encoder(lr))
    t_hrf, sup = T(lambda: net.hrfeat(fea))
    t_dec, hf = T(lambda: net.decoder1(*feats))
    t_reg, _ = T(lambda: net.reg(hf, sup))
t_fwd_grad, outs = T(lambda: net(lr, fea), 2)
def fb():
    o = net(lr, fea); (o[0].sum() + o[1].sum() + o[2].sum()).backward(); return o
t_fb, _ = T(fb, 2)
t_step, _ = T(lambda: ts(batch), 3)
print(f"B={B}: rrdb fwd {t_rrdb:.1f} | encoder fwd(no grad) {t_enc:.1f} | hrfeat fwd {t_hrf:.1f} | decoder fwd {t_dec:.1f} | reg fwd {t_reg:.1f} | model fwd(train, grad) {t_fwd_grad:.1f} | fwd+bwd {t_fb:.1f} | full step {t_step:.1f} ms")
# backward split: head only (detach encoder feats)
def fb_head():
    s = net.hrfeat(fea); h = net.reg(hf.detach().requires_grad_(True), s); h.sum().backward()
t_head, _ = T(fb_head, 2)
def fb_enc():
    f = net.encoder(lr); d = net.decoder1(*f); d.sum().backward()
t_encb, _ = T(fb_enc, 2)
print(f"hrfeat+reg fwd+bwd {t_head:.1f} ms | encoder+decoder1 fwd+bwd {t_encb:.1f} ms")
