"""GPU: end-to-end check of the training path (train.py:244-265 as harness.TrainStep): RRDBNet features (no grad) ->
SRRegress_Cls_feature forward -> the three adaptive losses -> backward through every libsrbh head kernel and the stock-op
encoder / decoders -> Adam.  On ONE fixed synthetic batch the loss has to fall; per-op gradient parity lives in
test_gpu_head.py / test_gpu_model.py, this catches what only shows when everything runs together."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_fixed_batch_loss_decreases():
    from oracle import synth
    from srbh_amd.harness import TrainStep, synthetic_batch
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    dev = "cuda:0"
    net_hr = RRDBNet(3, 3, num_block=2)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=2, seed=1337, mode="init"))
    torch.manual_seed(1337)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True,
                                chans_build=7)
    ts = TrainStep(net_hr.to(dev), net.to(dev), dev, lr=1e-3)
    batch = synthetic_batch(8, 1337, dev)
    losses = [float(ts(batch)[0]) for _ in range(40)]
    assert all(l == l and l < 1e9 for l in losses)
    assert losses[-1] < 0.75 * losses[0], (losses[0], losses[-1])


def test_graph_replayed_step_equals_eager_step(monkeypatch):
    """TrainStep(graph=True): three eager steps, then the whole step (features, forward, losses, backward, Adam with
    capturable=True) captured into ONE HIP graph and replayed.  Same batch sequence as the eager TrainStep: the loss curve
    must agree step by step (drop-connect -- the only RNG -- off), new batches must reach the static buffers, and the loss falls."""
    from oracle import synth
    from srbh_amd import encoders
    from srbh_amd.harness import TrainStep, synthetic_batch
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    monkeypatch.setattr(encoders, "DROP_CONNECT", 0.0)
    dev = "cuda:0"
    batches = [synthetic_batch(4, 100 + i % 2, dev) for i in range(10)]
    curves = []
    for graph in (False, True):
        net_hr = RRDBNet(3, 3, num_block=1)
        net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=1, seed=7, mode="init"))
        torch.manual_seed(11)
        net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
        ts = TrainStep(net_hr.to(dev), net.to(dev), dev, lr=1e-4, graph=graph)
        curves.append([float(ts(b)[0]) for b in batches])
        if graph:
            assert ts._graph is not None and ts.steps == len(batches)
    eager, replay = curves
    assert all(l == l for l in replay)
    # identical arithmetic, but training-mode BatchNorm statistics are atomics (order-dependent last bits) and ten Adam steps
    # amplify them: a relative bound per step
    for i, (a, b) in enumerate(zip(eager, replay)):
        assert abs(a - b) <= 2e-2 * abs(a), (i, a, b)
    assert replay[-2] < replay[0] and replay[-1] < replay[1]      # (the two alternating batches, each against its own first visit)
