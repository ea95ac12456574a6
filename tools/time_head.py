import sys, torch
sys.path.insert(0, '.')
from srbh_amd import synth
from srbh_amd.hrfuse import HRfeature, HRfuse_residual
dev = 'cuda:0'; B = 64
torch.manual_seed(0)
hf = HRfeature(64, 16, 16).to(dev).train()
reg = HRfuse_residual(16, 16, 16, 1, 4).to(dev).train()
seg = HRfuse_residual(16, 16, 16, 7, 4).to(dev).train()
fea = torch.randn(B, 64, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
lo = torch.randn(B, 16, 64, 64, device=dev, requires_grad=True)
for it in range(3):
    s = hf(fea); h = reg(lo, s); b = seg(lo, s)
    (h.sum() + b.sum()).backward()
torch.cuda.synchronize()
