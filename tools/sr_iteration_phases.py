"""developer aid: where one iteration of the SR-stage trainer (RealESRGAN.optimize_parameters, SR/rrdbnet_arch.py:538-592) spends its time: the phases
timed with a device synchronisation at each boundary (so the sum exceeds the free-running iteration).  python tools/sr_iteration_phases.py [B] [mode]"""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd import rrdbnet_autograd as RA
from srbh_amd.rrdbnet import RealESRGAN
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "fast"
dev = "cuda:0"
RA.set_train_precision(mode)
torch.manual_seed(3)
m = RealESRGAN(3, 3, num_block=23, device=dev, is_train=True)
g = torch.Generator().manual_seed(9)
gt = torch.nn.functional.interpolate(torch.rand((B, 3, 32, 32), generator=g), scale_factor=8, mode="bilinear").to(dev)
lq = torch.nn.functional.avg_pool2d(gt, 4)
T = {}
def tick(name, t0):
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
def iteration():
    t = time.perf_counter()
    m.feed_data({"lq": lq, "gt": gt}); t = tick("feed_data (USM sharpen)", t)
    m.net_d.requires_grad_(False); m.optimizer_g.zero_grad()
    out = m.net_g(m.lq); t = tick("generator forward", t)
    l = m.cri_pix(out, m.gt_usm) + m.cri_gan(m.net_d(out), True, is_disc=False); t = tick("pixel + GAN terms (discriminator forward, frozen)", t)
    l.backward(); t = tick("backward (discriminator data gradients + generator)", t)
    m.optimizer_g.step(); t = tick("optimizer_g.step (702 tensors)", t)
    m.net_d.requires_grad_(True); m.optimizer_d.zero_grad()
    for x, real in ((m.gt, True), (out.detach().clone(), False)):
        m.cri_gan(m.net_d(x), real, is_disc=True).backward()
    t = tick("discriminator: 2 x (forward + backward)", t)
    m.optimizer_d.step(); t = tick("optimizer_d.step", t)
    m.model_ema(decay=m.ema_decay); t = tick("EMA", t)
for _ in range(2): iteration()
T.clear()
N = 5
for _ in range(N): iteration()
tot = sum(T.values())
for k, v in T.items(): print(f"{v / N * 1e3:8.2f} ms  {k}")
print(f"{tot / N * 1e3:8.2f} ms  sum (with a sync per phase)")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    m.feed_data({"lq": lq, "gt": gt}); m.optimize_parameters()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / N * 1e3:8.2f} ms  optimize_parameters() free-running")
