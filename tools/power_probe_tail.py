"""developer aid: rocm-smi power / sclk while a ONE-RRDB network's forward_feature runs back to back (at 32 tiles its time is the three persistent
tail convs, csrc/srbh_ptail.hip); run once per library (SRBH_LIB_PATH) to compare kernel forms under the package power cap."""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srbh_amd import synth
from srbh_amd.rrdbnet import RRDBNet
sd = synth.rrdbnet_state_dict(num_block=1, seed=1337, mode="init")
net = RRDBNet(3, 3, num_block=1); net.load_state_dict(sd); net = net.cuda().eval()
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        p = re.findall(r"Power.*?:\s*([\d.]+)", o)
        c = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", o)
        samples.append((p[:1], c[:1]))
        time.sleep(0.25)
x = synth.tiles(32, 8, 64, seed=1337)[:, :3].contiguous().cuda()
with torch.no_grad():
    for _ in range(20): net.forward_feature(x)
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 8:
        for _ in range(200): net.forward_feature(x)
        torch.cuda.synchronize(); n += 200
    dt = time.perf_counter() - t0
stop[0] = True; th.join()
pw = [float(s[0][0]) for s in samples[4:] if s[0]]
ck = [float(s[1][0]) for s in samples[4:] if s[1]]
print("lib %s: %.4f ms per forward_feature (1 RRDB, 32 tiles)  power avg %.0f W  sclk avg %.0f MHz  (%d samples)"
      % (os.environ.get("SRBH_LIB_PATH", "in-tree"), dt / n * 1e3, sum(pw) / max(1, len(pw)), sum(ck) / max(1, len(ck)), len(pw)))
