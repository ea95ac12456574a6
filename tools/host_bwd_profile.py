"""developer probe: cProfile of the training step's host side INCLUDING backward: autograd's device thread is a C++ thread that cProfile does not
see, so the step runs under torch.autograd.set_multithreading_enabled(False) (backward in the calling thread).  Batch 2: the numbers are host
time.   python tools/host_bwd_profile.py [rows]"""
import sys, time, cProfile, pstats, torch
sys.path.insert(0, '.')
from srbh_amd import synth
from srbh_amd.harness import TrainStep, synthetic_batch
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.rrdbnet import RRDBNet
dev = 'cuda:0'
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 45
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337))
torch.manual_seed(0)
net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
ts = TrainStep(net_hr.to(dev), net.to(dev), dev, status_every=0)
batch = synthetic_batch(2, 1, dev)
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(6): ts(batch, next_batch=batch)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): ts(batch, next_batch=batch)
    pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(rows)
st.sort_stats("cumtime").print_stats(30)
