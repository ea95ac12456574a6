# NOTE (round 6): the -DP3_ABL ablation knob this script was written for was removed from csrc/srbh_ptrunk3_kernel.h with the other dead knobs; kept as the record of how profiles/r04c / r05y were produced.
"""developer aid: trunk launch time + package power + shader clock for ONE library build (SRBH_LIB_PATH: tools/build_variant.py
-DP3_ABL=... variants give WRONG results by construction; only time / power / clock are read).  Loops forward_feature(B=32) for
~5 s with rocm-smi sampled beside it.    usage: SRBH_LIB_PATH=build/variants/libsrbh_x.so python tools/power_ablate.py <label>"""
import ctypes, re, subprocess, sys, threading, time
sys.path.insert(0, '.')
import torch
from srbh_amd import _lib, synth
from srbh_amd.rrdbnet import RRDBNet
label = sys.argv[1] if len(sys.argv) > 1 else "base"
zero = len(sys.argv) > 2 and sys.argv[2] == "zeros"
net = RRDBNet(3, 3); net.load_state_dict(synth.rrdbnet_state_dict(seed=1337, mode="init")); net = net.cuda().eval()
x = synth.tiles(32, 8, 64, seed=1337)[:, :3].contiguous().cuda()
if zero:
    x.zero_()
L = _lib.lib()
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        p = re.findall(r"Power.*?:\s*([\d.]+)", o); c = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", o)
        if p and c:
            samples.append((float(p[0]), int(c[0])))
        time.sleep(0.3)
with torch.no_grad():
    for _ in range(20): net.forward_feature(x)
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 5:
        for _ in range(50): net.forward_feature(x)
        torch.cuda.synchronize()
    L.srbh_trunk_timing(1)
    acc = []
    for _ in range(30):
        net.forward_feature(x)
        ms = ctypes.c_float(0.0); L.srbh_trunk_last_ms(ctypes.byref(ms)); acc.append(ms.value)
    L.srbh_trunk_timing(0)
    stop[0] = True; th.join()
acc.sort()
s = samples[3:] or samples
print("%-22s trunk %.4f ms (median of 30)   power %.0f W   sclk %.0f MHz   (%d samples)" % (
    label + (" zeros" if zero else ""), acc[len(acc) // 2], sum(a for a, _ in s) / len(s), sum(b for _, b in s) / len(s), len(s)))
