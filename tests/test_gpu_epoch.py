"""GPU: the BASELINE configs[3] driver -- harness.train_epoch / bench.py --workload epoch (train.py:225-271: one pass over
data/datalist_globe_train_0.7.csv, drop_last) -- executed at N=1 against hand-driven TrainStep calls on the same device-drawn batches.
The two-rank form runs through tests/test_gpu_bench_ranks.py (bench.py launched as the driver launches it, gloo on one GPU)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _nets(dev, seed=11):
    from oracle import synth
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    net_hr = RRDBNet(3, 3, num_block=1)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=1, seed=7, mode="init"))
    torch.manual_seed(seed)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
    return net_hr.to(dev), net.to(dev)


def test_train_epoch_equals_hand_driven_steps(monkeypatch):
    """train_epoch(max_steps=3) == three TrainStep calls on the batches synthetic_batch_device draws from the same seeds: same
    step count, same tiles-seen arithmetic (drop_last), same loss after every step (drop-connect -- the only other RNG -- off)."""
    from srbh_amd import encoders
    from srbh_amd.harness import TrainStep, synthetic_batch_device, train_epoch
    monkeypatch.setattr(encoders, "DROP_CONNECT", 0.0)
    dev = torch.device("cuda:0")
    B, n_tiles = 4, 4 * 5 + 3                            # 23 tiles: 5 full batches, the ragged rest is dropped (train.py:97)
    net_hr, net = _nets(dev)
    ts = TrainStep(net_hr, net, dev, lr=1e-4, status_every=0)
    steps, tiles, loss = train_epoch(ts, n_tiles, B, 0, 1, dev, seed=1337, max_steps=3)
    assert (steps, tiles) == (3, 12) and ts.steps == 3
    net_hr2, net2 = _nets(dev)
    ts2 = TrainStep(net_hr2, net2, dev, lr=1e-4, status_every=0)
    gen = torch.Generator(device=dev)
    hand = []
    for i in range(3):
        gen.manual_seed(1337 + 7919 * 0 + 104729 * i)        # harness.train_epoch's seeding rule (rank 0)
        hand.append(float(ts2(synthetic_batch_device(B, gen, dev))[0]))
    # training-mode BatchNorm statistics are atomics (order-dependent last bits): relative bound, as in test_gpu_train_converges
    assert abs(float(loss) - hand[-1]) <= 2e-2 * abs(hand[-1]), (float(loss), hand)
    # without max_steps: 23 // 4 = 5 steps, 20 tiles
    steps, tiles, _ = train_epoch(ts, n_tiles, B, 0, 1, dev, seed=1)
    assert (steps, tiles) == (5, 20) and ts.steps == 8


def test_epoch_batches_differ_between_steps_and_ranks():
    """every step and every rank draws its own tiles (seed + 7919 rank + 104729 step): a DP pass must not train N copies of one shard"""
    from srbh_amd.harness import synthetic_batch_device
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    seen = []
    for rank in (0, 1):
        for step in (0, 1):
            gen.manual_seed(1337 + 7919 * rank + 104729 * step)
            b = synthetic_batch_device(2, gen, dev)
            assert b[0].shape == (2, 8, 64, 64) and b[1].shape == (2, 256, 256) and b[2].shape == (2, 64, 64)
            assert b[3].dtype == torch.int64 and int(b[3].max()) <= 6 and float(b[0].min()) >= 0 and float(b[0].max()) < 1
            seen.append(b[0].flatten()[:64].clone())
    for i in range(len(seen)):
        for j in range(i + 1, len(seen)):
            assert not torch.equal(seen[i], seen[j])
