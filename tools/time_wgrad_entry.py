"""developer aid: fused BasicBlock-entry weight gradient (srbh_hconv_wgrad_entry_b16) at the training step's shapes.
SRBH_WGRAD_ENTRY_FUSE=1 (default: chunk loop inside the tile walk) | 2 (chunk-outer kernel) | 0 (two separate calls).
python tools/time_wgrad_entry.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srbh_amd import hrfuse as H, hrfuse_autograd as HA
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
nhwc = lambda t: t.contiguous(memory_format=torch.channels_last)
for name, c0, c1, x16 in (("64 -> 16 (HRfeature entry, fp16 features)", 64, 0, True), ("16+16 -> 16 (reg / seg entry)", 16, 16, False)):
    x0 = nhwc(torch.randn((B, c0, 256, 256), device=dev))
    if x16:
        x0 = x0.half()
    srcs = [x0] + ([nhwc(torch.randn((B, c1, 256, 256), device=dev))] if c1 else [])
    g3 = nhwc(torch.randn((B, 16, 256, 256), device=dev) * 1e-3).bfloat16()
    g1 = nhwc(torch.randn((B, 16, 256, 256), device=dev) * 1e-3).bfloat16()
    with H.head_precision("f16"), torch.no_grad():
        for _ in range(3):
            HA.conv_wgrad_entry(srcs, g3, g1, 16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            HA.conv_wgrad_entry(srcs, g3, g1, 16)
        e1.record(); torch.cuda.synchronize()
    mb = (x0.numel() * x0.element_size() + sum(t.numel() * t.element_size() for t in srcs[1:]) + 2 * g3.numel() * 2) / 1e6
    us = e0.elapsed_time(e1) / 10 * 1e3
    print("FUSE=%s  %-44s %7.1f us  %6.0f MB  %5.0f GB/s" % (os.environ.get("SRBH_WGRAD_ENTRY_FUSE", "1"), name, us, mb, mb / us * 1e3 / 1e3))
