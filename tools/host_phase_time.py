"""developer aid: host ISSUE time of each phase of a training step (perf_counter, no synchronisation inside a step) next to
the step's wall time: shows whether the step is bound by the host queueing launches or by the kernels themselves.
usage (GPU box): python tools/host_phase_time.py [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from srbh_amd.rrdbnet import RRDBNet
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.harness import TrainStep, synthetic_batch
from srbh_amd import hrfuse as H

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
net_hr = RRDBNet(3, 3, num_block=23).to(dev)
net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7).to(dev)
ts = TrainStep(net_hr, net, dev)
batch = synthetic_batch(64, 1337, dev)
for _ in range(4):
    ts(batch)
torch.cuda.synchronize()
names = ["rrdb", "fwd", "loss", "zero", "bwd", "opt"]
acc = dict.fromkeys(names, 0.0)
t00 = time.perf_counter()
for _ in range(steps):
    lr, height, height_aggre, build, weight, weight_aggre = batch
    t = [time.perf_counter()]
    with torch.no_grad():
        hr_fea = net_hr.forward_feature(lr.index_select(1, ts._rgb_idx))
    t.append(time.perf_counter())
    hp, bp, hpa = ts.net(lr, hr_fea)
    t.append(time.perf_counter())
    loss = (ts.criterion[0](hp.squeeze(1), height, weight) + ts.criterion[1](hpa.squeeze(1), height_aggre, weight_aggre)
            + ts.criterion[2](bp, build, weight))
    t.append(time.perf_counter())
    ts.optimizer.zero_grad(set_to_none=True)
    t.append(time.perf_counter())
    loss.backward()
    t.append(time.perf_counter())
    ts.optimizer.step()
    t.append(time.perf_counter())
    for i, n in enumerate(names):
        acc[n] += t[i + 1] - t[i]
t_issue = time.perf_counter() - t00
torch.cuda.synchronize()
t_wall = time.perf_counter() - t00
print("host issue %.2f ms/step, wall %.2f ms/step" % (1e3 * t_issue / steps, 1e3 * t_wall / steps))
print("  ".join("%s %.2f" % (n, 1e3 * acc[n] / steps) for n in names))
if len(sys.argv) > 2:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(steps):
        ts(batch)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(35)
if len(sys.argv) > 3:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(steps):
            ts(batch)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=60, max_name_column_width=60))
