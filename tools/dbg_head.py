import sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
from srbh_amd import hrfuse as H
from oracle import srbh_oracle as O
DEV='cuda:0'
torch.manual_seed(0)
def chk(name, got, want):
    print(f"{name}: rel {O.rel_l2(got.cpu(), want):.3e} max {O.max_rel(got.cpu(), want):.3e}")
for cin, cout, ks, hw in ((16,16,3,12),(32,16,3,12),(32,16,1,12),(64,16,3,12),(16,64,3,12),(16,7,3,12),(16,16,3,70)):
    conv = torch.nn.Conv2d(cin, cout, ks, 1, ks//2, bias=True).to(DEV)
    x = torch.randn(2, cin, hw, hw)
    with torch.no_grad():
        y,_ = H.hconv([H.to_nhwc(x.to(DEV))], conv, H._PackedConv())
    chk(f"conv {cin}->{cout} k{ks} hw{hw}", y, F.conv2d(x, conv.weight.cpu(), conv.bias.cpu(), 1, ks//2))
# two sources
conv = torch.nn.Conv2d(32, 16, 3, 1, 1, bias=False).to(DEV)
a, b = torch.randn(2,16,12,12), torch.randn(2,16,12,12)
with torch.no_grad():
    y,_ = H.hconv([H.to_nhwc(a.to(DEV)), H.to_nhwc(b.to(DEV))], conv, H._PackedConv())
chk("conv cat", y, F.conv2d(torch.cat([a,b],1), conv.weight.cpu(), None, 1, 1))
# pre transform
conv = torch.nn.Conv2d(16, 16, 3, 1, 1, bias=False).to(DEV)
x = torch.randn(2,16,12,12); sc = torch.rand(16)+0.5; sh = torch.randn(16)*0.3
with torch.no_grad():
    y,_ = H.hconv([H.to_nhwc(x.to(DEV))], conv, H._PackedConv(), pre=(sc.to(DEV), sh.to(DEV), True))
chk("conv pre", y, F.conv2d(F.relu(x*sc[None,:,None,None]+sh[None,:,None,None]), conv.weight.cpu(), None, 1, 1))
# stats
with torch.no_grad():
    y, st = H.hconv([H.to_nhwc(x.to(DEV))], conv, H._PackedConv(), want_stats=True)
st = st.view(64, 2, 16).sum(0).cpu()
ref = F.conv2d(x, conv.weight.cpu(), None, 1, 1).double()
print("stats sum", float((st[0]-ref.sum((0,2,3))).abs().max()), "sq", float((st[1]-(ref*ref).sum((0,2,3))).abs().max()))
# bn_add_relu
a = torch.randn(2,16,12,12); idt = torch.randn(2,16,12,12)
s1,h1,s2,h2 = [torch.randn(16) for _ in range(4)]
out = H.bn_add_relu(H.to_nhwc(a.to(DEV)), s1.to(DEV), h1.to(DEV), H.to_nhwc(idt.to(DEV)), s2.to(DEV), h2.to(DEV))
v = lambda t: t[None,:,None,None]
chk("bn_add_relu", out, F.relu(a*v(s1)+v(h1)+idt*v(s2)+v(h2)))
out = H.bn_add_relu(H.to_nhwc(a.to(DEV)), s1.to(DEV), h1.to(DEV), H.to_nhwc(idt.to(DEV)))
chk("bn_add_relu id", out, F.relu(a*v(s1)+v(h1)+idt))
