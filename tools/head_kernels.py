"""developer aid / PMC target: the training step's dominant HEAD kernels on their own, 16-channel 256x256 maps at the step's batch
size -- hconv forward (+BN statistics), data gradient, weight gradient, BatchNorm backward (reduce, apply), bn_add_relu.
usage:  rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python tools/head_kernels.py [B] [reps]  (then tools/pmc_head.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srbh_amd import hrfuse as H, hrfuse_autograd as HA

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
H.set_head_precision(os.environ.get("SRBH_HEAD_PRECISION", "f16"))
conv = torch.nn.Conv2d(16, 16, 3, 1, 1, bias=False).to(dev)
bn = torch.nn.BatchNorm2d(16).to(dev).train()
x = H.to_nhwc(torch.randn(B, 16, 256, 256, device=dev))
g = H.to_nhwc(torch.randn(B, 16, 256, 256, device=dev))
pk, pg = H._PackedConv(), HA._PackedGrad()
for _ in range(reps):
    c, st = H.hconv([x], conv, pk, want_stats=True)
    s, h, m, i = H.bn_scale_shift(bn, st, B * 256 * 256, True)
    HA.conv_dgrad(g, conv.weight, pg)
    HA.conv_wgrad([x], None, g, 16, 3)
    HA.bn_backward(g, c, m, i, bn.weight, None, True)
    H.bn_add_relu(c, s, h, x)
torch.cuda.synchronize()
