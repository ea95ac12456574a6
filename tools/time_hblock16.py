"""developer aid: a plain eval BasicBlock of the fp16 chain at the predict path's size (B x 256 x 256 x 16): srbh_hblock16_eval against the two
hconv16 launches it replaces, device time by HIP events over graph-free back-to-back calls.  usage: time_hblock16.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srbh_amd import hrfuse as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda:0"
torch.manual_seed(0)
blk = H.BasicBlock(16, 16).to(dev).eval()
x = (torch.randn((B, 16, 256, 256), device=dev) * 0.7).to(torch.float16).contiguous(memory_format=torch.channels_last)
for out_h16 in (True, False):
    for fused in (False, True, False, True):
        H.HBLOCK16 = fused
        with torch.no_grad(), H.head_precision("f16"):
            for _ in range(5):
                blk.forward_nhwc([x], out_h16=out_h16)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                blk.forward_nhwc([x], out_h16=out_h16)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        px = B * 65536
        by = px * ((32 + 32 if out_h16 else 32 + 64) if fused else (64 + (96 if out_h16 else 128)))
        print(f"out_h16={out_h16} fused={fused}: {us:8.1f} us per block   {by / us / 1e6:6.2f} TB/s of its own algorithmic bytes   wpc={os.environ.get('SRBH_HBLOCK16_WPC', '3')} wgs={os.environ.get('SRBH_HBLOCK16_WGS', '768')}")
