"""developer aid: run ONE part of a tiled-prediction batch eagerly, alone on the device, for a rocprofv3 --kernel-trace
(kernel durations without the other stream's contention; summarise with tools/pw_trace.py <dir> "").
usage: predict_part_kernels.py <lr|hr|fuse> [batch=256] [reps=6]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from srbh_amd.harness import features_for_head

part = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
_, net_hr, model = bench._make_nets(argparse.Namespace(num_block=23), dev, False)
net_hr.eval(); model.eval()
x = torch.randn((batch, 8, 64, 64), device=dev) * 0.25 + 0.35
with torch.no_grad():
    fea = features_for_head(net_hr, x[:, :3], model=model)
    lr = model.forward_lr(x)
    hr = model.forward_hr(fea)
    torch.cuda.synchronize()
    for _ in range(reps):
        if part == "lr":
            model.forward_lr(x)
        elif part == "hr":
            model.forward_hr(fea)
        else:
            model.forward_fuse(lr[0], lr[1], hr)
        torch.cuda.synchronize()
