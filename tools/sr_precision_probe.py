"""Forward error of the SR generator's fp16-operand paths against the strict fp32 path on stress weights (2 blocks, 64 x 64 LR tiles):
inference trunk, and the "mixed" / "fast" training graphs.  python tools/sr_precision_probe.py (GPU)."""
import sys
sys.path.insert(0, '.')
import torch
from oracle import synth
from srbh_amd import rrdbnet_autograd as RA
from srbh_amd.rrdbnet import RRDBNet
sd = synth.rrdbnet_state_dict(num_block=2, seed=31, mode="stress")
g = torch.Generator().manual_seed(140)
x = torch.rand((2, 3, 64, 64), generator=g).cuda()
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
net = RRDBNet(3, 3, num_block=2); net.load_state_dict(sd); net = net.cuda().eval()
with torch.no_grad():
    y_inf, f_inf = net(x).float(), net.forward_feature(x).float()
    net.precision = "f32"
    y_ex, f_ex = net(x).float(), net.forward_feature(x).float()
    del net.precision
print("inference fp16 trunk vs strict: forward %.3e  forward_feature %.3e" % (rel(y_inf, y_ex), rel(f_inf, f_ex)))
for mode in ("mixed", "fast"):
    RA.set_train_precision(mode)
    n2 = RRDBNet(3, 3, num_block=2); n2.load_state_dict(sd); n2 = n2.cuda().train().enable_training_path(True)
    y = n2(x.clone().requires_grad_(True))
    yf = n2.forward_feature(x.clone().requires_grad_(True))
    print("fast workspace used:", bool(RA._FAST_WS)) if mode == "fast" else None
    print(mode, "training graph vs strict: forward %.3e  forward_feature %.3e" % (rel(y.detach(), y_ex), rel(yf.detach(), f_ex)))
RA.set_train_precision("f32")
