"""developer aid: soak the PIPELINED training step's feature prefetch -- the trunk in launches of 16 images on a second stream while the main
stream runs the encoder / decoder chains and the head's chip-filling kernels: every prefetched tensor must equal the inline forward_feature
of the same batch bit for bit (the in-launch halo exchange runs with other kernels taking and leaving CUs), error word clean.
python tools/soak_pipelined.py [steps] [compare every n-th]"""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd import synth, hrfuse as H
from srbh_amd.harness import TrainStep, synthetic_batch, features_for_head
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.rrdbnet import RRDBNet
dev = 'cuda:0'
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337))
torch.manual_seed(0)
net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
ts = TrainStep(net_hr.to(dev), net.to(dev), dev, status_every=0)
batches = [synthetic_batch(64, 10 + i, dev) for i in range(3)]
with torch.no_grad(), H.head_precision("f16"):
    want = [features_for_head(net_hr, b[0].index_select(1, ts._rgb_idx), True, model=net).clone() for b in batches]
torch.cuda.synchronize()
bad = 0
t0 = time.perf_counter()
every = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for i in range(steps):
    ts(batches[i % 3], next_batch=batches[(i + 1) % 3])
    if i % every == every - 1:
        got = ts._pf()                      # (waits for the trunk stream)
        w = want[(i + 1) % 3]
        if not torch.equal(got, w):
            bad += 1
            d = (got != w)
            idx = d.nonzero()
            imgs = sorted(set(idx[:, 0].tolist()))
            rows = idx[:, 2]
            print(f"  step {i}: {int(d.sum())} differing elements, images {imgs[:8]}{'...' if len(imgs) > 8 else ''}, output rows {int(rows.min())}..{int(rows.max())}, "
                  f"channels {int(idx[:, 1].min())}..{int(idx[:, 1].max())}, cols {int(idx[:, 3].min())}..{int(idx[:, 3].max())}, "
                  f"max |diff| {float((got.float() - w.float()).abs().max()):.3e}, NaN {bool(torch.isnan(got.float()).any())}", flush=True)
        net_hr.check_status()
torch.cuda.synchronize()
print(f"{steps} pipelined steps in {time.perf_counter() - t0:.1f} s ({ts.pipelined_steps} consumed prefetched features), {steps // every} prefetches compared bit for bit: {bad} mismatches, status ok")
