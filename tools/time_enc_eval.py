"""developer aid: EfficientNet-B4 encoder in eval mode, fused inference MBConv blocks (encoders.MBCONV_EVAL) vs the separate launches.
python tools/time_enc_eval.py [--eager]   (--eager: plain launches, for rocprofv3 --kernel-trace --stats; default: graph replays)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srbh_amd import encoders
eager = "--eager" in sys.argv
torch.manual_seed(0)
enc = encoders.get_encoder("efficientnet-b4", in_channels=8, depth=5, weights=None).cuda().eval()
for B in ((128,) if eager else (32, 128)):
    x = torch.rand((B, 8, 64, 64), device="cuda")
    for mode in ((encoders.MBCONV_EVAL,) if eager else (True, False, True, False)):
        encoders.MBCONV_EVAL = mode
        with torch.no_grad():
            for _ in range(3):
                enc(x)
            torch.cuda.synchronize()
            if eager:
                for _ in range(10):
                    enc(x)
                torch.cuda.synchronize()
                continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                enc(x)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record(); torch.cuda.synchronize()
        print("B", B, "fused" if mode else "separate", round(e0.elapsed_time(e1) / 20, 3), "ms (graph replay)")
