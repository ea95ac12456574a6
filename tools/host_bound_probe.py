"""developer probe: the HOST's time per training step.  At batch 2 the GPU work per launch is tiny, so the step time is (almost) what the host
needs to issue a step -- the floor under the batch-64 step whenever the GPU side gets faster.  Prints it for the serial and the pipelined
step, plus a cProfile of the issuing thread and a per-function profile of autograd's backward thread (threading.setprofile).
python tools/host_bound_probe.py"""
import sys, time, threading, cProfile, pstats, collections, torch
sys.path.insert(0, '.')
from srbh_amd import synth
from srbh_amd.harness import TrainStep, synthetic_batch
from srbh_amd.models import SRRegress_Cls_feature
from srbh_amd.rrdbnet import RRDBNet
dev = 'cuda:0'
net_hr = RRDBNet(3, 3); net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337))
torch.manual_seed(0)
net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
ts = TrainStep(net_hr.to(dev), net.to(dev), dev, status_every=0)
batch = synthetic_batch(2, 1, dev)
for pipe in (False, True):
    for _ in range(6): ts(batch, next_batch=batch if pipe else None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for _ in range(n): ts(batch, next_batch=batch if pipe else None)
    torch.cuda.synchronize()
    print(f"batch 2, {'pipelined' if pipe else 'serial'}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per step (~ host issue time)")
# profile: main thread by cProfile; every other thread (autograd's device thread) by a call-count / time table
acc = collections.defaultdict(lambda: [0, 0.0])
stack = threading.local()


def prof(frame, event, arg):
    if event == "call":
        st = getattr(stack, "s", None)
        if st is None: st = stack.s = []
        st.append((frame.f_code, time.perf_counter()))
    elif event == "return":
        st = getattr(stack, "s", None)
        if st:
            code, t0 = st.pop()
            a = acc[(code.co_filename.split('/')[-1], code.co_firstlineno, code.co_name)]
            a[0] += 1; a[1] += time.perf_counter() - t0


threading.setprofile(prof)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): ts(batch, next_batch=batch)
pr.disable(); torch.cuda.synchronize()
threading.setprofile(None)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
print("other threads (autograd's backward thread), cumulative time per function, 10 steps:")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {v[1] * 1e3:8.2f} ms  {v[0]:6d} calls  {k[0]}:{k[1]} {k[2]}")
