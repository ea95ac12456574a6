"""GPU: RRDBNet features handed to the head as an fp16 channels_last tensor (RRDBNet.forward_feature(out_dtype=float16),
harness.features_for_head; round-3 VERDICT What's weak #7).  conv_hr (SR/rrdbnet_arch.py:238) rounds ONCE in its epilogue -- the
rounding the head's fp16-operand entry kernel (SR/HRfuse.py:142-159: conv1 + downsample[0] of HRfeature's first BasicBlock) applied
while staging the fp32 tensor -- so every forward value of the fp16 head mode is unchanged, bit for bit."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = "cuda:0"


def _nets(num_block=1, seed=5):
    from oracle import synth
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    net_hr = RRDBNet(3, 3, num_block=num_block)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=num_block, seed=seed, mode="stress"))
    torch.manual_seed(seed)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
    return net_hr.to(DEV).eval(), net.to(DEV)


@pytest.mark.parametrize("B", [1, 3, 5])          # (1, 3: fewer than 512 tiles -- the NHWC16 store exists in the persistent tail kernel only)
def test_feature_h16_is_the_fp32_feature_rounded_once(B):
    from oracle import synth
    net_hr, _ = _nets()
    x = synth.tiles(B, 8, 64, seed=3)[:, :3].contiguous().to(DEV)
    with torch.no_grad():
        y32 = net_hr.forward_feature(x)
        y16 = net_hr.forward_feature(x, out_dtype=torch.float16)
    assert y16.dtype == torch.float16 and y16.shape == y32.shape and y16.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y16, y32.half())
    net_hr.check_status()
    with pytest.raises(ValueError):
        net_hr.forward_feature(x, out=torch.empty_like(y32), out_dtype=torch.float16)
    net_hr.precision = "f32"
    with pytest.raises(ValueError):
        net_hr.forward_feature(x, out_dtype=torch.float16)


def test_eval_outputs_do_not_change_by_a_bit():
    from oracle import synth
    from srbh_amd import harness
    from srbh_amd import hrfuse as H
    net_hr, net = _nets()
    net.eval()
    x = synth.tiles(4, 8, 64, seed=9).to(DEV)
    with torch.no_grad():
        assert H.head_h16()                                  # 'auto' + no_grad: the fp16-operand inference chain
        f16 = harness.features_for_head(net_hr, x[:, :3].contiguous())
        assert f16.dtype == torch.float16
        f32 = net_hr.forward_feature(x[:, :3].contiguous())
        net(x, f32)                       # warm-up: the stock-op decoders' first call at a shape runs MIOpen's solver search, which may
        torch.cuda.synchronize()          # execute (and return the result of) another algorithm than every later call
        b = net(x, f32)
        a = net(x, f16)
        b2 = net(x, f32)
    for u, v in zip(b, b2):
        assert torch.equal(u, v)          # (the comparison below means something only if the fp32 hand-off repeats itself)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # the exact-fp32 head refuses an fp16 tensor instead of silently widening it
    with H.head_precision("f32"), torch.no_grad(), pytest.raises(TypeError):
        net(x, f16)


def test_entry_kernel_and_weight_gradients_from_fp16_features():
    """The training-mode entry (conv1 3x3 + downsample 1x1 over the 64-channel features, with BatchNorm statistics): outputs and
    statistics from the fp16 tensor are bit-identical to those from the fp32 tensor it was rounded from; the two entry weight
    gradients (bf16 operands) see fp16 -> bf16 instead of fp32 -> bf16: a double rounding inside a bf16 product."""
    from srbh_amd import hrfuse as H
    from srbh_amd import hrfuse_autograd as HA
    _, net = _nets()
    blk = net.hrfeat[0]
    g = torch.Generator().manual_seed(4)
    x = torch.randn((3, 64, 64, 128), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    x16 = x.half()
    dy = (torch.randn((3, 16, 64, 128), generator=g) * 1e-3).to(DEV).contiguous(memory_format=torch.channels_last)
    with H.head_precision("f16"), torch.no_grad():
        c1, s1, d, sd = H.hconv_entry([x], blk.conv1, blk._p1, blk.downsample[0], blk._pd, want_stats=True)
        c2, s2, d2, sd2 = H.hconv_entry([x16], blk.conv1, blk._p1, blk.downsample[0], blk._pd, want_stats=True)
        assert torch.equal(c1, c2) and torch.equal(d, d2) and torch.equal(s1, s2) and torch.equal(sd, sd2)
        for ks in (3, 1):
            for gy in (dy, dy.bfloat16()):
                a = HA.conv_wgrad([x16], None, gy, 16, ks).double()
                b = HA.conv_wgrad([x], None, gy, 16, ks).double()
                assert float((a - b).norm() / b.norm()) < 2e-3, (ks, gy.dtype)
                ref = HA.conv_wgrad([x16.float().contiguous(memory_format=torch.channels_last)], None, gy, 16, ks).double()
                assert float((a - ref).norm() / ref.norm()) < 1e-6, (ks, gy.dtype)       # (fp16 -> bf16 of the SAME values: same products)


def test_train_step_runs_on_fp16_features(monkeypatch):
    """TrainStep(head_precision='f16') end to end on the fp16 hand-off: same loss as with the fp32 hand-off up to the step's own
    run-to-run noise (training-mode BatchNorm statistics are atomics: two runs of ONE configuration already differ by ~4e-3 in the
    height maps), weights stay finite, the exact-fp32 mode keeps taking fp32 features."""
    from srbh_amd import encoders, harness
    monkeypatch.setattr(encoders, "DROP_CONNECT", 0.0)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(harness, "FEATURE_H16", mode)
        net_hr, net = _nets()
        seen = []
        orig = net.hrfeat.forward
        net.hrfeat.forward = lambda x, out_h16=False, _o=orig, _s=seen: (_s.append(x.dtype), _o(x, out_h16))[1]
        ts = harness.TrainStep(net_hr, net, DEV, lr=1e-4, status_every=0)
        batch = harness.synthetic_batch(4, 21, DEV)
        losses = [float(ts(batch)[0]) for _ in range(3)]
        assert seen == [torch.float16 if mode else torch.float32] * 3
        assert all(torch.isfinite(p).all() for p in net.parameters())
        res[mode] = losses
    for a, b in zip(res[True], res[False]):
        assert abs(a - b) <= 2e-2 * abs(b), (res[True], res[False])
    net_hr, net = _nets()
    seen = []
    orig = net.hrfeat.forward
    net.hrfeat.forward = lambda x, out_h16=False, _o=orig, _s=seen: (_s.append(x.dtype), _o(x, out_h16))[1]
    harness.TrainStep(net_hr, net, DEV, lr=1e-4, status_every=0, head_precision="f32")(harness.synthetic_batch(2, 3, DEV))
    assert seen == [torch.float32]


def test_no_head_entry_point_falls_back_to_its_slow_form(monkeypatch):
    """srbh_path_counters (round 4): the head entry points that choose between a specialised kernel and the template -- or one fused pass
    and two launches -- count their choice.  With the shapes of the model (64-channel fp16 features, 16 + 16 fp32 fuse entries) NO block
    entry may run split, neither forward (eval chain and training) nor in the weight gradients; the fp16 feature hand-off had silently
    split the inference chain's 64-channel entry until a kernel trace showed it."""
    from oracle import synth
    from srbh_amd import _lib, encoders, harness
    monkeypatch.setattr(encoders, "DROP_CONNECT", 0.0)
    net_hr, net = _nets()
    x = synth.tiles(2, 8, 64, seed=9).to(DEV)
    net.eval()
    with torch.no_grad():
        net(x, harness.features_for_head(net_hr, x[:, :3].contiguous()))       # warm-up (packs)
        _lib.path_counters(reset=True)
        net(x, harness.features_for_head(net_hr, x[:, :3].contiguous()))
    c = _lib.path_counters(reset=True)
    assert c["entry_fused"] == 3 and c["entry_split"] == 0, c                  # hrfeat.0, reg.fuse.0, seg.fuse.0
    assert c["hblock16"] == 6, c                                               # the 6 plain blocks: one pass each (round 6, srbh_hblock16_eval)
    assert c["hconv16"] >= 5, c                                                # conv2 of the 3 entry blocks + the two conv_last
    assert c["hconv_up"] == 4, c                                               # the two Upsampler convs of reg and seg: counted, not silent
    ts = harness.TrainStep(net_hr, net, DEV, lr=1e-4, status_every=0)
    batch = harness.synthetic_batch(2, 3, DEV)
    ts(batch)
    _lib.path_counters(reset=True)
    ts(batch)
    c = _lib.path_counters(reset=True)
    assert c["entry_fused"] == 3 and c["entry_split"] == 0, c
    assert c["wgrad_entry_fused"] == 3 and c["wgrad_entry_split"] == 0 and c["wgrad_f32"] == 0, c
    # the 16 -> 16 weight gradients: inside srbh_hbwd16 (conv2 of all 9 blocks + conv1 of the 6 plain ones), the narrow conv_last ones on wgrad16
    assert c["hbwd16"] == 15 and c["wgrad16"] >= 3, c


def test_fp16_hrfeat_and_upsampler_hand_offs_do_not_change_a_bit(monkeypatch):
    """Round 4: inside the inference chain HRfeature's output and the Upsampler's PixelShuffle outputs (SR/HRfuse.py:17-44,173-190) are
    written as fp16 NHWC tensors -- each value rounded once, where the consuming conv's staging rounded the fp32 tensor -- so that both
    sources of the reg / seg entry blocks are fp16 and go through the fused fp16 entry kernel: every output equals the fp32 hand-off's,
    bit for bit; the Upsampler alone: fp16 result == the fp32 result rounded once."""
    from oracle import synth
    from srbh_amd import _lib, harness, models
    from srbh_amd import hrfuse as H
    net_hr, net = _nets()
    net.eval()
    x = synth.tiles(4, 8, 64, seed=13).to(DEV)
    with torch.no_grad():
        f16 = harness.features_for_head(net_hr, x[:, :3].contiguous())
        net(x, f16)
        torch.cuda.synchronize()
        monkeypatch.setattr(models, "HRFEAT_OUT_H16", False)
        b = net(x, f16)
        monkeypatch.setattr(models, "HRFEAT_OUT_H16", True)
        _lib.path_counters(reset=True)
        a = net(x, f16)
        paths = _lib.path_counters()
        for u, v in zip(a, b):
            assert u.dtype == torch.float32 and torch.equal(u, v)
        assert paths["entry_fused"] >= 3 and paths["entry_split"] == 0, paths      # hrfeat's entry and both fuse entries: ONE launch each
        lr16 = torch.randn((3, 16, 64, 64), device=DEV)
        up = net.reg.upsampler
        y32, y16 = up(lr16), up(lr16, out_h16=True)
        assert y16.dtype == torch.float16 and y16.shape == (3, 16, 256, 256) and torch.equal(y16, y32.half())


@pytest.mark.parametrize("B,Hh,Ww,stats,o16", [(2, 8, 128, True, False), (3, 12, 64, False, True), (5, 256, 256, True, False), (1, 4, 64, False, False)])
def test_whole_row_entry_kernel_equals_the_chunked_entry(B, Hh, Ww, stats, o16):
    """Round 4: HRfeature's entry (SR/HRfuse.py:142-159, 64 fp16 channels) on hconv_entry64_kernel -- 16-byte loads, 8 lanes per 128-byte
    pixel row, all four chunks of a tile staged at once -- against the chunked fused kernel fed with the SAME values as an fp32 tensor
    (which it rounds to the same halves): both outputs bit-identical (fp32 and fp16 outputs, with and without the BatchNorm statistics),
    the statistics equal up to their order of addition; 5 x 256 x 256 = 1 280 tiles on 256 workgroups walks several tiles per workgroup."""
    import torch.nn as nn
    from srbh_amd import _lib
    from srbh_amd import hrfuse as H
    g = torch.Generator().manual_seed(B * 100 + Ww)
    x16 = (torch.randn((B, 64, Hh, Ww), generator=g) * 0.7).to(DEV).half().contiguous(memory_format=torch.channels_last)
    x32 = x16.float().contiguous(memory_format=torch.channels_last)
    conv1, convd = nn.Conv2d(64, 16, 3, 1, 1, bias=False).to(DEV), nn.Conv2d(64, 16, 1, bias=False).to(DEV)
    post = ((torch.rand(16, generator=g) + 0.5).to(DEV), torch.randn(16, generator=g).to(DEV))
    with H.head_precision("f16"), torch.no_grad():
        kw = dict(want_stats=stats, out_h16=o16) if stats or o16 else {}
        if not stats:
            kw.update(post1=post, post1_relu=True, postd=post)
        _lib.path_counters(reset=True)
        a = H.hconv_entry([x16], conv1, H._PackedConv(), convd, H._PackedConv(), **kw)
        b = H.hconv_entry([x32], conv1, H._PackedConv(), convd, H._PackedConv(), **kw)
        assert _lib.path_counters()["entry_fused"] == 2
    assert a[0].dtype == (torch.float16 if o16 else torch.float32)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and float(a[0].float().abs().max()) > 0
    if stats:
        fold = lambda t: t.view(-1, 2, 16).sum(0)
        for u, v in ((a[1], b[1]), (a[3], b[3])):
            # (fp32 partial sums per workgroup over other tile sets; bounded against the largest moment, not element-wise)
            assert float((fold(u) - fold(v)).abs().max()) <= 1e-5 * float(fold(v).abs().max())
