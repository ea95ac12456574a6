"""developer probe: does the frozen RRDBNet trunk of the NEXT batch run beside the rest of the training step (encoder / decoders:
chains of small kernels; head: HBM-bound kernels) when it is launched on a second stream with fewer workgroups than CUs?
The trunk kernel takes one CU per workgroup (all 160 KiB of LDS): 32 images per launch fill the chip and nothing else gets in.
With `sub` images per launch (sub * 8 workgroups) the other CUs stay free for the other stream.
Prints ms per step: serial (as TrainStep does), and overlapped for several `sub`.   python tools/partition_probe.py [B]"""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd import synth, hrfuse as H
from srbh_amd.harness import TrainStep, synthetic_batch, features_for_head
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.rrdbnet import RRDBNet
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337))
torch.manual_seed(0)
net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
ts = TrainStep(net_hr.to(dev), net.to(dev), dev)
batch = synthetic_batch(B, 1, dev)
for _ in range(4): ts(batch)
lr, height, height_aggre, build, weight, weight_aggre = batch
x3 = lr.index_select(1, ts._rgb_idx).contiguous()


def trunk(sub):
    with torch.no_grad(), H.head_precision("f16"):
        return [features_for_head(net_hr, x3[i:i + sub], True, model=net) for i in range(0, B, sub)]


def rest(hr_fea):
    with H.head_precision("f16"):
        hp, bp, hpa = ts.net(lr, hr_fea)
        loss = (ts.criterion[0](hp.squeeze(1), height, weight) + ts.criterion[1](hpa.squeeze(1), height_aggre, weight_aggre)
                + ts.criterion[2](bp, build, weight))
        ts.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        ts.optimizer.step()


fea = torch.cat(trunk(32)) if B > 32 else trunk(B)[0]
torch.cuda.synchronize()


def T(fn, n=30):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print(f"B={B}: trunk alone: " + " | ".join(f"sub {s}: {T(lambda: trunk(s)):.2f} ms" for s in (32, 24, 16, 12, 8)))
print(f"rest of the step alone (features given): {T(lambda: rest(fea)):.2f} ms")
print(f"serial trunk(32) + rest: {T(lambda: (trunk(32), rest(fea))):.2f} ms")
sT = torch.cuda.Stream()
main = torch.cuda.current_stream()
for sub in (32, 24, 16, 12, 8):
    def both():
        sT.wait_stream(main)
        with torch.cuda.stream(sT):
            keep = trunk(sub)
        rest(fea)
        main.wait_stream(sT)
        return keep
    print(f"overlapped, trunk on a second stream with {sub} images ({sub * 8} workgroups) per launch: {T(both):.2f} ms per step")

# ---- pipelined order: the small-kernel chains contiguous (encoder + decoders forward first, their backward last), the NEXT batch's trunk
# launched on the second stream when backward leaves the big head kernels (post-accumulate hook on hrfeat's first weight)
print("pipelined order: encoder / decoders first, trunk(next) launched when hrfeat's backward is done")
m = ts.net
state = {"sub": 0, "keep": None}
first_w = m.hrfeat[0].conv1.weight


def _hook(_p):
    if state["sub"]:
        ev = torch.cuda.Event(); ev.record(main)
        sT.wait_event(ev)
        with torch.cuda.stream(sT):
            state["keep"] = trunk(state["sub"])


first_w.register_post_accumulate_grad_hook(_hook)


def rest_lr_first(hr_fea):
    with H.head_precision("f16"), H.defer_batch_counters():
        enc = m.encoder(lr)
        hf = m.decoder1(*enc)
        ha = m._aggre(hf)
        bf = m.decoder2(*enc)
        main.wait_stream(sT)                 # the features of THIS batch (launched during the previous step)
        sup = m.hrfeat(hr_fea)
        hp = m.reg(hf, sup)
        bp = m.seg(bf, sup)
        loss = (ts.criterion[0](hp.squeeze(1), height, weight) + ts.criterion[1](ha.squeeze(1), height_aggre, weight_aggre)
                + ts.criterion[2](bp, build, weight))
    ts.optimizer.zero_grad(set_to_none=True)
    with H.head_precision("f16"):
        loss.backward()
    ts.optimizer.step()


for rnd in range(3):
    state["sub"] = 0
    line = [f"serial trunk(32)+rest(lr first) {T(lambda: (trunk(32), rest_lr_first(fea))):.2f}", f"serial trunk(32)+rest(as shipped) {T(lambda: (trunk(32), rest(fea))):.2f}"]
    for sub in (32, 24, 20, 16, 12):
        state["sub"] = sub
        line.append(f"pipelined sub {sub}: {T(lambda: rest_lr_first(fea)):.2f}")
    state["sub"] = 0
    print(f"round {rnd}: " + " | ".join(line))
