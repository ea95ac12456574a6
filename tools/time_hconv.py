"""developer timing of single head-conv shapes (B=64, 256x256): forward conv, data gradient, weight gradient."""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd import hrfuse as H, hrfuse_autograd as HA
dev = 'cuda:0'; B = 64
def T(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for cin, cout in ((16, 16), (32, 16), (64, 16)):
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=False).to(dev)
    x = H.to_nhwc(torch.randn(B, cin, 256, 256, device=dev))
    g = H.to_nhwc(torch.randn(B, cout, 256, 256, device=dev))
    pk, pg = H._PackedConv(), HA._PackedGrad()
    t_f = T(lambda: H.hconv([x], conv, pk, want_stats=True))
    t_f0 = T(lambda: H.hconv([x], conv, pk, want_stats=False))
    t_d = T(lambda: HA.conv_dgrad(g, conv.weight, pg))
    t_w = T(lambda: HA.conv_wgrad([x], None, g, cout, 3))
    gf = 2 * 9 * cin * cout * B * 65536 / 1e9
    print(f"{cin}->{cout}: fwd+stats {t_f:.0f} us | fwd {t_f0:.0f} us ({gf / t_f0 * 1e-3:.0f} TF) | dgrad {t_d:.0f} us | wgrad {t_w:.0f} us ({gf / t_w * 1e-3:.0f} TF) | {gf:.1f} GF, fp32-MFMA floor {gf / 157e3 * 1e6:.0f} us, HBM floor fwd {(cin + cout) * 4 * B * 65536 / 5e12 * 1e6:.0f} us")
