import os, sys, torch
sys.path.insert(0, '.')
from oracle import synth
from srbh_amd.rrdbnet import RRDBNet
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 23
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sd = synth.rrdbnet_state_dict(num_block=nb, seed=1337, mode="init")
net = RRDBNet(3, 3, num_block=nb); net.load_state_dict(sd); net = net.cuda().eval()
x = synth.tiles(B, 8, 64, seed=1337)[:, :3].contiguous().cuda()
with torch.no_grad():
    os.environ["SRBH_PERSISTENT"] = "0"; y0 = net.forward_feature(x).clone()
    os.environ["SRBH_PERSISTENT"] = "1"
    for rep in range(3):
        y1 = net.forward_feature(x); 
        try: net.check_status(); st = "ok"
        except Exception as e: st = str(e)
        bad = ~torch.isfinite(y1)
        diff = (y1 != y0)
        per_img = diff.flatten(1).any(1).nonzero().flatten().tolist()
        rows = diff[per_img[0]].any(0).any(1).nonzero().flatten().tolist() if per_img else []
        print(f"rep {rep}: status {st}; nonfinite {int(bad.sum())}; mismatching images {per_img[:40]}; rows of first: {rows[:6]}..{rows[-3:] if rows else ''}; y0 finite {bool(torch.isfinite(y0).all())}")
