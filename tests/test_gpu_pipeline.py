"""GPU: the pipelined training step (harness.TrainStep(batch, next_batch=...)): the frozen RRDBNet's features of the
NEXT batch are computed on a second stream, in launches that leave CUs free, beside this step's encoder / decoder phases
(train.py:244-246 computes them inline at the top of each step; the values are the same function of the same batch either way)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _nets(dev, seed=11, blocks=1):
    from oracle import synth
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    net_hr = RRDBNet(3, 3, num_block=blocks)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=blocks, seed=7, mode="init"))
    torch.manual_seed(seed)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
    return net_hr.to(dev), net.to(dev)


def test_prefetched_features_are_the_inline_features_bit_for_bit(monkeypatch):
    """several launches into batch slices of ONE tensor on the second stream == one inline forward_feature(out_dtype=float16)"""
    from srbh_amd import hrfuse as H
    from srbh_amd.harness import TrainStep, features_for_head, synthetic_batch
    monkeypatch.setenv("SRBH_PIPE_IMAGES", "3")          # 8 images -> launches of 3, 3, 2
    dev = "cuda:0"
    net_hr, net = _nets(dev, blocks=2)
    ts = TrainStep(net_hr, net, dev, status_every=0)
    batch = synthetic_batch(8, 5, dev)
    pf = ts._launch_prefetch(batch)
    got = pf()
    with torch.no_grad(), H.head_precision("f16"):
        want = features_for_head(net_hr, batch[0].index_select(1, ts._rgb_idx), True, model=net)
    torch.cuda.synchronize()
    net_hr.check_status()
    assert got.dtype == torch.float16 and want.dtype == torch.float16 and got.shape == want.shape
    assert torch.equal(got, want)
    assert pf.matches(batch[0]) and not pf.matches(synthetic_batch(8, 6, dev)[0])


def test_pipelined_steps_follow_the_serial_steps(monkeypatch):
    """the same batch sequence through the serial and the pipelined step: same loss curve (up to the BatchNorm atomics' last bits,
    amplified by Adam: the bound of the graph-replay test), every step but the first consumed prefetched features, and a batch that was
    NOT the announced one is computed inline instead of being served the wrong features"""
    from srbh_amd import encoders
    from srbh_amd.harness import TrainStep, synthetic_batch
    monkeypatch.setattr(encoders, "DROP_CONNECT", 0.0)
    monkeypatch.setenv("SRBH_PIPE_IMAGES", "3")
    dev = "cuda:0"
    batches = [synthetic_batch(4, 100 + i % 3, dev) for i in range(8)]
    curves = []
    for pipelined in (False, True):
        net_hr, net = _nets(dev)
        ts = TrainStep(net_hr, net, dev, lr=1e-4, status_every=0)
        cur = []
        for i, b in enumerate(batches):
            nxt = batches[i + 1] if (pipelined and i + 1 < len(batches)) else None
            cur.append(float(ts(b, next_batch=nxt)[0]))
        curves.append(cur)
        torch.cuda.synchronize()
        net_hr.check_status()
        assert ts.pipelined_steps == (len(batches) - 1 if pipelined else 0)
    serial, piped = curves
    assert all(v == v for v in piped)
    for i, (a, b) in enumerate(zip(serial, piped)):
        assert abs(a - b) <= 2e-2 * abs(a), (i, a, b)
    # announce one batch, bring another: the prefetch must not be used
    net_hr, net = _nets(dev)
    ts = TrainStep(net_hr, net, dev, lr=1e-4, status_every=0)
    ts(batches[0], next_batch=batches[1])
    l_other = float(ts(batches[2])[0])
    assert ts.pipelined_steps == 0 and l_other == l_other
    net_hr2, net2 = _nets(dev)
    ts2 = TrainStep(net_hr2, net2, dev, lr=1e-4, status_every=0)
    ts2(batches[0])
    l_ref = float(ts2(batches[2])[0])
    assert abs(l_other - l_ref) <= 2e-2 * abs(l_ref)


def test_pipelined_step_with_the_exact_fp32_head_prefetches_fp32_features(monkeypatch):
    """head_precision="f32" (the mode the <= 5e-5 gradient-parity tests pin): the prefetch hands over fp32 features, bit-identical to the inline
    forward_feature, and the pipelined loss curve follows the serial one"""
    from srbh_amd import encoders
    from srbh_amd.harness import TrainStep, synthetic_batch
    monkeypatch.setattr(encoders, "DROP_CONNECT", 0.0)
    monkeypatch.setenv("SRBH_PIPE_IMAGES", "3")
    dev = "cuda:0"
    batches = [synthetic_batch(4, 200 + i % 2, dev) for i in range(5)]
    curves = []
    for pipelined in (False, True):
        net_hr, net = _nets(dev)
        ts = TrainStep(net_hr, net, dev, lr=1e-4, status_every=0, head_precision="f32")
        cur = []
        for i, b in enumerate(batches):
            nxt = batches[i + 1] if (pipelined and i + 1 < len(batches)) else None
            cur.append(float(ts(b, next_batch=nxt)[0]))
            if pipelined and nxt is not None:
                assert ts._pf is not None and ts._pf.fea.dtype == torch.float32
                with torch.no_grad():
                    want = net_hr.forward_feature(nxt[0].index_select(1, ts._rgb_idx))
                assert torch.equal(ts._pf(), want)
        curves.append(cur)
        assert ts.pipelined_steps == (len(batches) - 1 if pipelined else 0)
    for i, (a, b) in enumerate(zip(*curves)):
        assert abs(a - b) <= 2e-2 * abs(a), (i, a, b)


def test_prefetch_beside_a_running_step_is_bit_identical_full_depth():
    """the full 23-RRDB trunk in 16-image launches on the second stream WHILE the step's kernels take and leave CUs: the contended
    regime in which a neighbour's progress word really is behind when a wave polls it (tools/soak_pipelined.py is the long form; it
    found the one-short vmcnt window of the per-wave poll, srbh_ptrunk3_kernel.h `stage_item`).  Every second prefetch is compared
    bit for bit with the inline features of the same batch."""
    from oracle import synth
    from srbh_amd import hrfuse as H
    from srbh_amd.harness import TrainStep, features_for_head, synthetic_batch
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    dev = "cuda:0"
    net_hr = RRDBNet(3, 3)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(seed=1337))
    torch.manual_seed(0)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
    ts = TrainStep(net_hr.to(dev), net.to(dev), dev, status_every=0)
    batches = [synthetic_batch(64, 10 + i, dev) for i in range(2)]
    with torch.no_grad(), H.head_precision("f16"):
        want = [features_for_head(net_hr, b[0].index_select(1, ts._rgb_idx), True, model=net).clone() for b in batches]
    torch.cuda.synchronize()
    for i in range(80):
        ts(batches[i % 2], next_batch=batches[(i + 1) % 2])
        if i % 2:
            assert torch.equal(ts._pf(), want[(i + 1) % 2]), f"prefetched features of step {i} differ from the inline forward"
    torch.cuda.synchronize()
    net_hr.check_status()
    assert ts.pipelined_steps == 79


def test_a_new_batch_at_a_recycled_address_is_not_served_the_old_prefetch():
    """round-5 ADVICE: the announced batch is dropped by the caller and ANOTHER batch of the same shape is allocated -- the caching
    allocator hands out the same address with version 0.  The prefetch holds the announced tensor itself (identity, not address), so
    the new batch is computed inline; the abandoned launches are waited for before the inline forward touches the shared workspace."""
    from srbh_amd.harness import TrainStep, synthetic_batch
    dev = "cuda:0"
    net_hr, net = _nets(dev, blocks=2)
    ts = TrainStep(net_hr, net, dev, lr=1e-4, status_every=0)
    b0 = synthetic_batch(4, 300, dev)
    b1 = list(synthetic_batch(4, 301, dev))
    ts(b0, next_batch=tuple(b1))
    assert ts._pf is not None
    other = synthetic_batch(4, 302, dev)
    stale = ts._pf
    lr_new = torch.empty_like(b1[0])
    lr_new.copy_(other[0])
    assert not stale.matches(lr_new)
    # same storage, same version, same shape as the announced tensor but another Python object standing for "a recycled address"
    alias = b1[0].view_as(b1[0])
    assert alias.data_ptr() == b1[0].data_ptr() and not stale.matches(alias)
    loss = float(ts((lr_new,) + tuple(other[1:]))[0])
    assert ts.pipelined_steps == 0 and loss == loss
    torch.cuda.synchronize()
    net_hr.check_status()
