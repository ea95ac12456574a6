# NOTE (round 6): the -DP3_ABL ablation knob this script was written for was removed from csrc/srbh_ptrunk3_kernel.h with the other dead knobs; kept as the record of how profiles/r04c / r05y were produced.
"""developer aid: trunk launch time (HIP events inside libsrbh) without any correctness check -- for ABLATED variants (tools/build_variant.py
-DP3_ABL=..., wrong results by construction) loaded through SRBH_LIB_PATH.  usage: time_trunk.py [reps]"""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from srbh_amd import _lib, synth
from srbh_amd.rrdbnet import RRDBNet
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
net = RRDBNet(3, 3); net.load_state_dict(synth.rrdbnet_state_dict(seed=1337, mode="init")); net = net.cuda().eval()
x = synth.tiles(32, 8, 64, seed=1337)[:, :3].contiguous().cuda()
L = _lib.lib()
with torch.no_grad():
    for _ in range(5): net.forward_feature(x)
    L.srbh_trunk_timing(1)
    acc = []
    for _ in range(reps):
        net.forward_feature(x)
        ms = ctypes.c_float(0.0); L.srbh_trunk_last_ms(ctypes.byref(ms)); acc.append(ms.value)
    L.srbh_trunk_timing(0)
acc.sort()
print("trunk ms: median %.4f  min %.4f  mean %.4f" % (acc[len(acc) // 2], acc[0], sum(acc) / len(acc)))
