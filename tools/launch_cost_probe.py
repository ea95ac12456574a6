"""developer probe: what ONE more small launch costs inside the training step.  The step is timed as it is and with N extra launches
per step appended to the same stream -- (a) 4-byte fills (the floor of a dispatch), (b) a dependent chain of small libsrbh kernels'
stand-in: torch adds on a 64 KB tensor, each reading the previous one's output.  d(step) / N is what removing a launch of that kind
returns.  python tools/launch_cost_probe.py [B]"""
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
for _ in range(5): ts(batch)
tiny = torch.zeros(1, device=dev)
chain = torch.zeros(16384, device=dev)


def run(n_fill, n_chain, steps=12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        ts(batch)
        for _ in range(n_fill): tiny.zero_()
        x = chain
        for _ in range(n_chain): x = x + 1.0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rnd in range(2):
    base = run(0, 0)
    for n in (100, 400):
        f = run(n, 0); c = run(0, n)
        print(f"round {rnd}: step {base:.3f} ms | +{n} fills {f:.3f} ms ({(f - base) / n * 1e3:.2f} us each) | +{n} chained adds {c:.3f} ms ({(c - base) / n * 1e3:.2f} us each)")
