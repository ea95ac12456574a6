"""developer aid: forward / gradient error of the mixed-precision head against the exact-fp32 graph for the TRAIN_IO16 variants"""
import sys, torch
sys.path.insert(0, '.')
from oracle import srbh_oracle as O
from srbh_amd import hrfuse as H
from srbh_amd.hrfuse import HRfeature, HRfuse_residual
DEV = "cuda:0"
def rnd(shape, seed): return torch.randn(shape, generator=torch.Generator().manual_seed(seed))
def run(mode, io16, act):
    H.set_head_precision(mode); H.TRAIN_IO16 = io16; H.TRAIN_IO16_ACT = act
    torch.manual_seed(5)
    hf, fu = HRfeature(64, 16, 16).to(DEV).train(), HRfuse_residual(16, 16, 16, 1, 4).to(DEV).train()
    x = rnd((4, 64, 128, 128), 11).to(DEV); lo = rnd((4, 16, 32, 32), 12).to(DEV).requires_grad_(True)
    y = fu(lo, hf(x)); (y.square().mean() * 1e-3).backward()
    return y.detach().cpu(), {k: p.grad.cpu() for k, p in list(hf.named_parameters()) + list(fu.named_parameters())}, lo.grad.cpu()
ref = run("f32", False, "none")
def cos(a, b): return float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()).clamp_min(1e-300))
for io16, act in ((False, "none"), (True, "none"), (True, "c1"), (True, "c2"), (True, "c1c2")):
    r = run("f16", io16, act)
    rels = sorted(O.rel_l2(r[1][k], g) for k, g in ref[1].items() if float(g.norm()) > 1e-3 * max(float(v.norm()) for v in ref[1].values()))
    cs = min(cos(r[1][k], g) for k, g in ref[1].items() if float(g.norm()) > 1e-3 * max(float(v.norm()) for v in ref[1].values()))
    print(f"io16={io16} act={act:5s}: out rel {O.rel_l2(r[0], ref[0]):.2e}  grad median rel {rels[len(rels)//2]:.2e} max {rels[-1]:.2e} min cos {cs:.4f}  dlo rel {O.rel_l2(r[2], ref[2]):.2e}")
H.set_head_precision("auto")
