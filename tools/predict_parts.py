"""developer aid: where a batch of the tiled prediction spends its time.  Times the three HIP graphs of harness._PredictGraph
(encoder / decoders `g_lr`, RRDBNet features + HRfeature `g_hr`, reg / seg `g_fuse`) each ALONE, the trunk alone, and the batch as
predict_tiles replays it (g_lr on the second stream beside g_hr, then g_fuse): the sum of the parts against the whole says how much
of the encoder's time is hidden.   usage: predict_parts.py [batch=256] [reps=20]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench                                            # (registers the package under its import name)
from srbh_amd import harness
from srbh_amd.harness import _PredictGraph

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
args = argparse.Namespace(num_block=23)
torch.backends.cudnn.benchmark = True
_, net_hr, model = bench._make_nets(args, dev, False)
net_hr.eval(); model.eval()
harness.PREDICT_AHEAD = False
with torch.no_grad():
    pg = _PredictGraph(net_hr, model, batch, dev, 8)
    x = torch.randn((batch, 8, 64, 64), device=dev) * 0.25 + 0.35
    pg.x.copy_(x)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def whole():
    cur = torch.cuda.current_stream(dev)
    pg.side.wait_stream(cur)
    with torch.cuda.stream(pg.side):
        pg.g_lr.replay()
    pg.g_hr.replay()
    cur.wait_stream(pg.side)
    pg.g_fuse.replay()


def serial():
    pg.g_lr.replay(); pg.g_hr.replay(); pg.g_fuse.replay()


def trunk():
    with torch.no_grad():
        net_hr.forward_feature(pg.x[:, :3])


for rnd in range(2):
    t = {"g_lr": timed(pg.g_lr.replay), "g_hr": timed(pg.g_hr.replay), "g_fuse": timed(pg.g_fuse.replay), "features(eager)": timed(trunk),
         "serial": timed(serial), "whole": timed(whole)}
    parts = t["g_lr"] + t["g_hr"] + t["g_fuse"]
    print("batch %d  " % batch + "  ".join("%s %.3f" % kv for kv in t.items()) + "  | parts %.3f  hidden %.3f ms  -> %.0f tiles/s (graphs only)"
          % (parts, parts - t["whole"], batch / t["whole"] * 1e3), flush=True)

# ---- the arrangement predict_tiles uses (harness.PREDICT_AHEAD): batch k + 1's encoder / decoders behind batch k's trunk
harness.PREDICT_AHEAD = True
with torch.no_grad():
    pg2 = _PredictGraph(net_hr, model, batch, dev, 8)


def ahead():
    pg2(x, batch, x)


for rnd in range(3):
    print("whole (encoder beside this batch's trunk) %.3f ms   ahead (next batch's encoder behind the trunk) %.3f ms   trunk graph %.3f  hrfeat graph %.3f"
          % (timed(whole), timed(ahead), timed(pg2.g_trunk.replay), timed(pg2.g_hrfeat.replay)), flush=True)
pg2.reset()
