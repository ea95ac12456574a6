"""developer aid: soak the persistent trunk -- many forwards at several batch sizes, error word checked, output must be
bit-stable from call to call (the in-launch halo exchange has no tolerance for races)."""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd import synth
from srbh_amd.rrdbnet import RRDBNet
dev = 'cuda:0'
net = RRDBNet(3, 3); net.load_state_dict(synth.rrdbnet_state_dict(seed=1337, mode="init")); net = net.to(dev).eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for B in (32, 5, 33, 64, 1):
    x = synth.tiles(B, 8, 64, seed=7 + B)[:, :3].contiguous().to(dev)
    with torch.no_grad():
        ref = net.forward_feature(x).clone()
        t0 = time.perf_counter(); bad = 0
        reps = n if B == 32 else n // 5
        for i in range(reps):
            y = net.forward_feature(x)
            if i % 50 == 49:
                net.check_status()
                if not torch.equal(y, ref): bad += 1
        torch.cuda.synchronize()
    print(f"B={B}: {reps} forwards in {time.perf_counter() - t0:.1f} s, mismatching checks {bad}, status ok", flush=True)
