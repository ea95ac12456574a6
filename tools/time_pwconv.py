"""developer aid: the 1x1 convolutions of EfficientNet-B4's MBConv blocks (expand / gated project with folded BatchNorm and skip) at the
tiled prediction's batch, device time per launch by HIP events; run once per library (SRBH_PW_LDS=0: pw_gemm_kernel only).
usage: time_pwconv.py [B=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srbh_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = _lib.lib()
dev = "cuda:0"
# (Cin, Cexp, Cout, HW_in, HW_out, repeats) per stage of B4 at a 64x64 tile: first block of the stage, then its repeats
STAGES = [(24, 144, 32, 1024, 256, 1), (32, 192, 32, 256, 256, 3), (32, 192, 56, 256, 64, 1), (56, 336, 56, 64, 64, 3), (56, 336, 112, 64, 16, 1),
          (112, 672, 112, 16, 16, 5), (112, 672, 160, 16, 16, 1), (160, 960, 160, 16, 16, 5), (160, 960, 272, 16, 4, 1), (272, 1632, 272, 4, 4, 7),
          (272, 1632, 448, 4, 4, 1), (448, 2688, 448, 4, 4, 1)]
tot = 0.0
flops = 0.0
for cin, cexp, cout, hwi, hwo, rep in STAGES:
    for name, K, M, HW, epi in (("expand", cin, cexp, hwi, 0), ("project", cexp, cout, hwo, 1)):
        x = torch.randn((B, K, HW), device=dev)
        wt = (torch.randn((K, M), device=dev) / K ** 0.5)
        y = torch.empty((B, M, HW), device=dev)
        gate, sc, sh = torch.rand((B, K), device=dev), torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev)
        res = torch.randn((B, M, HW), device=dev)
        def run():
            if epi:
                _lib.check(L.srbh_pwconv_fwd_epi(x.data_ptr(), wt.data_ptr(), 1, y.data_ptr(), B, K, M, HW, gate.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                 res.data_ptr(), 0, _lib.stream_ptr()), "epi")
            else:
                _lib.check(L.srbh_pwconv_fwd_wt(x.data_ptr(), wt.data_ptr(), y.data_ptr(), B, K, M, HW, _lib.stream_ptr()), "wt")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * M * K * HW * B
        tot += us * rep
        flops += fl * rep
        print("%-7s M %4d K %4d HW %4d x%d  %7.1f us  %6.1f TF/s" % (name, M, K, HW, rep, us, fl / us / 1e6), flush=True)
print("all 1x1 convs of one encoder pass at B = %d: %.3f ms, %.1f TF/s" % (B, tot / 1e3, flops / tot / 1e6))
