"""developer aid: every 16-bit-IO kernel form against its fp32-IO form on inputs that are exactly representable in the 16-bit type"""
import sys, torch
sys.path.insert(0, '.')
from oracle import srbh_oracle as O
from srbh_amd import hrfuse as H, hrfuse_autograd as HA
DEV = "cuda:0"
torch.manual_seed(0)
B, C, Hh, Ww = 2, 16, 64, 128
def nh(t): return H.to_nhwc(t)
def rb(t): return t.to(torch.bfloat16).float()
def rh(t): return t.to(torch.float16).float()
def as16(t, dt):   # NHWC 16-bit tensor with the same values
    o = H.empty_nhwc(*t.shape, t.device, dt); o.copy_(t); return o
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))
H.set_head_precision("f16")
g = nh(rb(torch.randn(B, C, Hh, Ww, device=DEV) * 1e-5)); c = nh(rh(torch.randn(B, C, Hh, Ww, device=DEV)))
out = nh(torch.randn(B, C, Hh, Ww, device=DEV))
mean = torch.randn(C, device=DEV) * 0.1; invstd = torch.rand(C, device=DEV) + 0.5; gamma = torch.rand(C, device=DEV) + 0.5
s1 = torch.rand(C, device=DEV) + 0.5; h1 = torch.randn(C, device=DEV) * 0.1
# BN backward: relu form
r32 = HA.bn_backward(g, c, mean, invstd, gamma, None, True, relu_ref=out)
r16 = HA.bn_backward(g, as16(c, torch.float16), mean, invstd, gamma, None, True, relu_ref=out, out_b16=True)
print("bn relu: dc", rel(r16[0], r32[0]), "dgamma", rel(r16[1], r32[1]), "dbeta", rel(r16[2], r32[2]), "dz", rel(r16[3], r32[3]), r16[0].dtype, r16[3].dtype)
# BN backward: masked form with bf16 g
r32 = HA.bn_backward(g, c, mean, invstd, gamma, (s1, h1), True)
r16 = HA.bn_backward(as16(g, torch.bfloat16), as16(c, torch.float16), mean, invstd, gamma, (s1, h1), True, out_b16=True)
print("bn mask: dc", rel(r16[0], r32[0]), "dgamma", rel(r16[1], r32[1]), "dbeta", rel(r16[2], r32[2]))
# dgrad 16->16
conv = torch.nn.Conv2d(16, 16, 3, 1, 1, bias=False).to(DEV)
res = nh(rb(torch.randn(B, C, Hh, Ww, device=DEV) * 1e-5))
d32 = HA.conv_dgrad(g, conv.weight, HA._PackedGrad(), res=res)
d16 = HA.conv_dgrad(as16(g, torch.bfloat16), conv.weight, HA._PackedGrad(), res=as16(res, torch.bfloat16))
d16b = HA.conv_dgrad(as16(g, torch.bfloat16), conv.weight, HA._PackedGrad(), res=as16(res, torch.bfloat16), out_b16=True)
print("dgrad16: fp32 out", rel(d16, d32), "bf16 out", rel(d16b, d32), d16b.dtype)
# dgrad entry shapes: 16 -> 64 (3x3) and 16 -> 32 1x1
for cin, ks in ((64, 3), (32, 3), (32, 1), (64, 1)):
    cv = torch.nn.Conv2d(cin, 16, ks, 1, ks // 2, bias=False).to(DEV)
    rs = nh(rb(torch.randn(B, cin, Hh, Ww, device=DEV) * 1e-5))
    a32 = HA.conv_dgrad(g, cv.weight, HA._PackedGrad(), res=rs)
    a16 = HA.conv_dgrad(as16(g, torch.bfloat16), cv.weight, HA._PackedGrad(), res=as16(rs, torch.bfloat16))
    a16b = HA.conv_dgrad(as16(g, torch.bfloat16), cv.weight, HA._PackedGrad(), out_b16=True)
    a32n = HA.conv_dgrad(g, cv.weight, HA._PackedGrad())
    print(f"dgrad 16->{cin} k{ks}: res16", rel(a16, a32), "out16", rel(a16b, a32n))
# wgrad 16->16: x fp16 with pre, dy bf16
x = c
w32 = HA.conv_wgrad([x], (s1, h1, True), g, 16, 3)
w16 = HA.conv_wgrad([as16(x, torch.float16)], (s1, h1, True), as16(g, torch.bfloat16), 16, 3)
w16d = HA.conv_wgrad([x], (s1, h1, True), as16(g, torch.bfloat16), 16, 3)
w16x = HA.conv_wgrad([as16(x, torch.float16)], None, g, 16, 3); w32x = HA.conv_wgrad([x], None, g, 16, 3)
print("wgrad16: both", rel(w16, w32), "dy only", rel(w16d, w32), "x only", rel(w16x, w32x))
for cin, ks in ((64, 3), (32, 3), (32, 1), (64, 1)):
    xs = nh(torch.randn(B, cin, Hh, Ww, device=DEV))
    a32 = HA.conv_wgrad([xs], None, g, 16, ks); a16 = HA.conv_wgrad([xs], None, as16(g, torch.bfloat16), 16, ks)
    print(f"wgrad {cin}->16 k{ks}: dy16", rel(a16, a32))
# bn_add_relu
sa = torch.rand(C, device=DEV) + 0.5; ha = torch.randn(C, device=DEV) * 0.1
o32 = H.bn_add_relu(c, sa, ha, out); o16 = H.bn_add_relu(as16(c, torch.float16), sa, ha, out)
o32d = H.bn_add_relu(c, sa, ha, c, s1, h1); o16d = H.bn_add_relu(as16(c, torch.float16), sa, ha, as16(c, torch.float16), s1, h1)
print("bn_add_relu:", rel(o16, o32), rel(o16d, o32d))
H.set_head_precision("auto")
