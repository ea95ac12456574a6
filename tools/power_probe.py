"""developer aid: sample rocm-smi power / sclk while the trunk runs back to back (real vs all-zero activations)."""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, '.')
import torch
from srbh_amd import synth
from srbh_amd.rrdbnet import RRDBNet
sd = synth.rrdbnet_state_dict(seed=1337, mode="init")
net = RRDBNet(3, 3); net.load_state_dict(sd); net = net.cuda().eval()
samples = []
stop = [False]
def sampler():
    while not stop[0]:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
            p = re.findall(r"Power.*?:\s*([\d.]+)", o)
            c = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", o)
            t = re.findall(r"Temperature \(Sensor (?:junction|edge)\) \(C\):\s*([\d.]+)", o)
            samples.append((time.perf_counter(), p[:1], c[:1], t[:1]))
        except Exception as e:
            samples.append((time.perf_counter(), str(e)))
        time.sleep(0.25)
for kind in ("idle", "real", "zeros", "real_b16"):
    B = 16 if kind == "real_b16" else 32
    x = synth.tiles(B, 8, 64, seed=1337)[:, :3].contiguous().cuda()
    if kind == "zeros": x.zero_()
    samples.clear(); stop[0] = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    with torch.no_grad():
        if kind == "idle":
            time.sleep(3)
        else:
            while time.perf_counter() - t0 < 6:
                for _ in range(50): net.forward_feature(x)
                torch.cuda.synchronize(); n += 50
    dt = time.perf_counter() - t0
    stop[0] = True; th.join()
    print(kind, "ms/forward", dt / max(n, 1) * 1e3, "samples:", [s[1:] for s in samples[2::3]][:8])
o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
print(o[-1500:])
