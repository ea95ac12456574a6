"""developer aid: device time of the decoder 3x3 convolutions (csrc/srbh_dconv.hip) at the ten shapes of one U-Net decoder, forward /
data gradient / weight gradient, next to the stock convolution (MIOpen).  HIP events around 20 back-to-back calls.
usage: python tools/time_dconv.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from srbh_amd import _lib, encoders as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.backends.cudnn.benchmark = len(sys.argv) > 2 and sys.argv[2] == "find"       # MIOpen solver search, as bench_predict runs it
dev = "cuda:0"
L = _lib.lib()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0] * 6
for Cin, Cout, W in [(608, 256, 4), (256, 256, 4), (312, 128, 8), (128, 128, 8), (160, 64, 16), (64, 64, 16), (112, 32, 32), (32, 32, 32), (32, 16, 64), (16, 16, 64)]:
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1, bias=False).to(dev)
    x = torch.randn(B, Cin, W, W, device=dev)
    gy = torch.randn(B, Cout, W, W, device=dev)
    packs = E._DecoderConvPacks()
    pf, pb = packs.fwd(conv.weight), packs.bwd(conv.weight)
    y, dx, dw = torch.empty_like(gy), torch.empty_like(x), torch.empty_like(conv.weight)
    ws = torch.empty(L.srbh_dconv_wgrad_ws_floats(B, Cin, Cout, W, W), dtype=torch.float32, device=dev)
    st = _lib.stream_ptr()
    t = [timeit(lambda: L.srbh_dconv_fwd(x.data_ptr(), pf.data_ptr(), y.data_ptr(), B, Cin, Cout, W, W, 0, st)),
         timeit(lambda: L.srbh_dconv_fwd(gy.data_ptr(), pb.data_ptr(), dx.data_ptr(), B, Cout, Cin, W, W, 1, st)),
         timeit(lambda: L.srbh_dconv_wgrad(x.data_ptr(), gy.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, Cin, Cout, W, W, st))]
    with torch.no_grad():
        t.append(timeit(lambda: F.conv2d(x, conv.weight, None, 1, 1)))
        t.append(timeit(lambda: torch.ops.aten.convolution_backward(gy, x, conv.weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])))
        t.append(timeit(lambda: torch.ops.aten.convolution_backward(gy, x, conv.weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])))
    gf = 2 * 9 * Cin * Cout * B * W * W / 1e9
    print("%4d->%3d @%2dx%-2d %5.2f GF | libsrbh fwd %6.1f  dgrad %6.1f  wgrad %6.1f us (%5.1f / %5.1f / %5.1f TF/s) | stock fwd %6.1f  dgrad %6.1f  wgrad %6.1f us"
          % (Cin, Cout, W, W, gf, t[0], t[1], t[2], gf / t[0] * 1e3, gf / t[1] * 1e3, gf / t[2] * 1e3, t[3], t[4], t[5]))
    for i in range(6):
        tot[i] += t[i]
print("one decoder, B=%d: libsrbh fwd %.0f + dgrad %.0f + wgrad %.0f = %.0f us | stock %.0f + %.0f + %.0f = %.0f us"
      % (B, tot[0], tot[1], tot[2], sum(tot[:3]), tot[3], tot[4], tot[5], sum(tot[3:])))
