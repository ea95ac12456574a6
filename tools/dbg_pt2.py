import os, sys, torch
sys.path.insert(0, '.')
from oracle import synth
from srbh_amd.rrdbnet import RRDBNet
nb = int(sys.argv[1]); B = int(sys.argv[2])
sd = synth.rrdbnet_state_dict(num_block=nb, seed=1337, mode="init")
net = RRDBNet(3, 3, num_block=nb); net.load_state_dict(sd); net = net.cuda().eval()
x = synth.tiles(B, 8, 64, seed=1337)[:, :3].contiguous().cuda()
with torch.no_grad():
    os.environ["SRBH_PERSISTENT"] = "0"; y0 = net.forward_feature(x).clone()
    os.environ["SRBH_PERSISTENT"] = "1"
    for rep in range(2):
        y1 = net.forward_feature(x); net.check_status()
        d = (y1 - y0).abs()
        rel = float(d.max() / y0.abs().max())
        bad_img = (d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()
        i = bad_img[0] if bad_img else 0
        rows = (d[i].amax((0, 2)) > 0).nonzero().flatten().tolist()
        cols = (d[i].amax((0, 1)) > 0).nonzero().flatten().tolist()
        print(f"rep{rep} nb={nb} B={B}: max rel diff {rel:.3e}; bad images {len(bad_img)}; img {i}: rows {rows[:4]}..{rows[-2:]} ({len(rows)}) cols ({len(cols)}) {cols[:3]}")
