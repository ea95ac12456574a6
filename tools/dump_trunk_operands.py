"""Dump REAL trunk operands for tools/mfma_ceiling: fp16 activations (conv_first output and X1..X4 of the first dense block, NHWC, as
the MFMA B operand sees them) and fp16 weights (the five convs of body.0.rdb1) of the synthetic 'init' network on real synthetic tiles.
Computed with stock torch ops on the device (values, not fragment layout, are what the power measurement needs).
    python tools/dump_trunk_operands.py gpurun_out/trunk_acts.bin gpurun_out/trunk_weights.bin"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srbh_amd import synth  # noqa: E402

sd = synth.rrdbnet_state_dict(num_block=23, seed=1337, mode="init")
dev = "cuda:0"
x = synth.tiles(4, 8, 64, seed=1337)[:, :3].contiguous().to(dev)
g = lambda k: sd[k].to(dev)                                     # noqa: E731
feat = F.conv2d(x, g("conv_first.weight"), g("conv_first.bias"), 1, 1)
planes, ws = [feat], []
for k in range(1, 6):
    w, b = g(f"body.0.rdb1.conv{k}.weight"), g(f"body.0.rdb1.conv{k}.bias")
    ws.append(w.half().flatten())
    y = F.conv2d(torch.cat(planes, 1), w, b, 1, 1)
    if k < 5:
        planes.append(F.leaky_relu(y, 0.2))
acts = torch.cat([p.permute(0, 2, 3, 1).contiguous().half().flatten() for p in planes])
acts.cpu().numpy().tofile(sys.argv[1])
torch.cat(ws).cpu().numpy().tofile(sys.argv[2])
print("acts", acts.numel(), "rms", float(acts.float().pow(2).mean().sqrt()), "weights", torch.cat(ws).numel(), "rms", float(torch.cat(ws).float().pow(2).mean().sqrt()))
