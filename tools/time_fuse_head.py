"""developer aid: HRfuse_residual (Upsampler + 3 BasicBlocks + conv_last) in the inference chain at B tiles, with the HR features handed
over as fp16 NHWC vs fp32; per-kernel durations via torch events around each module.  python tools/time_fuse_head.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srbh_amd import hrfuse as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda:0"
torch.manual_seed(0)
m = H.HRfuse_residual(hr_chans=16, lr_chans=16, mid_chans=16, out_chans=1, upscale=4).to(dev).eval()
x_lr = torch.randn((B, 16, 64, 64), device=dev)
x_hr32 = torch.randn((B, 16, 256, 256), device=dev).contiguous(memory_format=torch.channels_last)
x_hr16 = x_hr32.half()


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for name, xh in (("fp32 x_hr", x_hr32), ("fp16 x_hr", x_hr16)):
        h16 = xh.dtype == torch.float16
        up = lambda: m.upsampler(x_lr, out_h16=h16)
        xl = up()
        blk0 = lambda: m.fuse[0].forward_nhwc([xl, xh], True)
        print("%s: whole %.3f ms | upsampler %.3f | entry block %.3f" % (name, t(lambda: m(x_lr, xh)), t(up), t(blk0)))
