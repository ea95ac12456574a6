"""developer aid: the full training step on ONE fixed synthetic batch must reduce the loss (end-to-end check of forward,
backward, optimizer beyond the per-op gradient parity tests)."""
import sys, torch
sys.path.insert(0, '.')
from srbh_amd import synth
from srbh_amd.harness import TrainStep, synthetic_batch
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.rrdbnet import RRDBNet
dev = 'cuda:0'
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337, mode="init"))
torch.manual_seed(1337)
net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
ts = TrainStep(net_hr.to(dev), net.to(dev), dev, lr=1e-3)
batch = synthetic_batch(16, 1337, dev)
losses = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    loss, _ = ts(batch)
    losses.append(float(loss))
    if i % 10 == 0: print(i, round(losses[-1], 3), flush=True)
print("first", round(losses[0], 3), "last", round(losses[-1], 3), "min", round(min(losses), 3), "finite", all(l == l for l in losses))
assert losses[-1] < 0.7 * losses[0], "loss did not decrease"
