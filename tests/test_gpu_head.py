"""GPU parity of the HR feature / fusion head (libsrbh fp32 matrix-core kernels) against the CPU oracle and the
fixtures captured from the imported reference.  The head is fp32 end to end: tolerance 2e-5 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import srbh_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _strict_head_precision():
    """These tests pin the STRICT mode of the head (exact-fp32 matrix cores, <= 2e-5 / 2e-4 against the reference); the default
    inference mode (fp16 operands) has its own tests with its own stated tolerance in tests/test_gpu_head_f16.py."""
    from srbh_amd import hrfuse
    hrfuse.set_head_precision("f32")
    yield
    hrfuse.set_head_precision("auto")


DEV = "cuda:0"
TOL = 2e-5


def rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def gold(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def test_packed_weights_follow_a_fused_optimizer():
    """a fused Adam step does not bump `weight._version`; the pack cache must still repack (wcache.py)"""
    from srbh_amd import hrfuse as H
    conv = torch.nn.Conv2d(16, 16, 3, 1, 1, bias=True).to(DEV)
    x = torch.randn(1, 16, 12, 20, device=DEV)
    pk = H._PackedConv()
    opt = torch.optim.Adam(conv.parameters(), lr=0.05, fused=True)
    for _ in range(2):
        with torch.no_grad():
            got, _ = H.hconv([H.to_nhwc(x)], conv, pk)
            want = torch.nn.functional.conv2d(x, conv.weight, conv.bias, 1, 1)
        assert O.rel_l2(got.cpu(), want.cpu()) <= 2e-5
        for p in conv.parameters():
            p.grad = torch.ones_like(p)
        opt.step()


def test_pixelshuffle_fold_bit_exact():
    """Upsampler with a channel-identity centre tap: the output must be the exact PixelShuffle gather."""
    from srbh_amd.hrfuse import Upsampler
    up = Upsampler(scale=2, n_feats=16)
    w = torch.zeros(64, 16, 3, 3)
    # conv output channel o copies input channel o % 16 scaled by (1 + o//16): values stay exact in fp32
    for o in range(64):
        w[o, o % 16, 1, 1] = 1.0 + o // 16
    up.load_state_dict({"0.weight": w, "0.bias": torch.zeros(64)})
    up = up.to(DEV).eval()
    x = torch.randint(-100, 100, (2, 16, 9, 70)).float()
    with torch.no_grad():
        y = up(x.to(DEV)).cpu()
    conv = torch.nn.functional.conv2d(x, w, None, 1, 1)
    assert torch.equal(y, O.pixel_shuffle(conv, 2))
    assert torch.equal(y, torch.nn.PixelShuffle(2)(conv))


@pytest.mark.parametrize("inp,planes,seed,tag,xseed", [(32, 16, 16, "g6_bb32_16", 106), (16, 16, 17, "g6_bb16_16", 107)])
def test_basicblock_eval_and_train_forward(inp, planes, seed, tag, xseed, golden_dir):
    from srbh_amd.hrfuse import BasicBlock
    g = gold(golden_dir, "g6_basicblock")
    sd = {}
    synth.basicblock_state_dict(sd, "", inp, planes, seed, "stress")
    blk = BasicBlock(inp, planes)
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(DEV)
    x = rnd((2, inp, 12, 12), xseed)
    with torch.no_grad():
        ye = blk.eval()(x.to(DEV)).cpu()
        assert O.rel_l2(ye, g[tag + "_eval"]) <= TOL
        assert O.rel_l2(ye, O.basic_block(synth.clone_sd(sd), "", x, False)) <= TOL
        yt = blk.train()(x.to(DEV)).cpu()
    assert O.rel_l2(yt, g[tag + "_train"]) <= TOL
    msd = blk.state_dict()
    for k, v in g.items():
        if k.startswith(tag + "_stat_"):
            name = k[len(tag + "_stat_"):]
            assert torch.allclose(msd[name].cpu().double(), v.double(), rtol=1e-5, atol=1e-6), name


def test_hrfeature_and_fuse_heads(golden_dir):
    from srbh_amd.hrfuse import HRfeature, HRfuse_residual
    g = gold(golden_dir, "g7_head")
    sd = synth.hrfeature_state_dict(64, 16, 16, seed=18, mode="stress")
    m = HRfeature(64, 16, 16)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    x = rnd((2, 64, 16, 16), 108)
    with torch.no_grad():
        assert O.rel_l2(m.eval()(x.to(DEV)).cpu(), g["g7_hrfeat_eval"]) <= TOL
        assert O.rel_l2(m.train()(x.to(DEV)).cpu(), g["g7_hrfeat_train"]) <= TOL
    for oc in (1, 7):
        sd = synth.hrfuse_residual_state_dict(16, 16, 16, oc, 4, seed=19 + oc, mode="stress")
        m = HRfuse_residual(16, 16, 16, oc, 4)
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV)
        a, b = rnd((2, 16, 4, 4), 109), rnd((2, 16, 16, 16), 110)
        with torch.no_grad():
            ye = m.eval()(a.to(DEV), b.to(DEV))
            assert ye.shape == (2, oc, 16, 16)
            assert O.rel_l2(ye.cpu(), g[f"g7_fuse{oc}_eval"]) <= TOL
            assert O.rel_l2(m.train()(a.to(DEV), b.to(DEV)).cpu(), g[f"g7_fuse{oc}_train"]) <= TOL


def test_head_full_resolution_ragged_and_variants():
    """256x256 (several tiles per image) and a ragged size; sibling modules against the oracle building blocks."""
    from srbh_amd.hrfuse import HRfeature, HRupsample, GeoNet, Refine_residual
    sd = synth.hrfeature_state_dict(64, 16, 16, seed=3, mode="stress")
    m = HRfeature(64, 16, 16)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    for shape in ((1, 64, 256, 256), (2, 64, 37, 91)):
        x = rnd(shape, 5)
        with torch.no_grad():
            y = m(x.to(DEV).contiguous(memory_format=torch.channels_last))
        assert O.rel_l2(y.cpu(), O.hrfeature(synth.clone_sd(sd), "", x, False)) <= TOL
    sdg = synth.hrfeature_state_dict(4, 16, 16, seed=4, mode="stress", prefix="feat.")
    gnet = GeoNet(4, 16)
    gnet.load_state_dict(sdg)
    x = rnd((1, 4, 20, 20), 6)
    with torch.no_grad():
        y = gnet.to(DEV).eval()(x.to(DEV))
    assert O.rel_l2(y.cpu(), O.hrfeature(sdg, "feat.", x, False)) <= TOL
    sdr = synth.hrfuse_residual_state_dict(16, 16, 16, 3, 4, seed=8, mode="stress")
    ref = Refine_residual(16, 16, 16, 3)
    ref.load_state_dict({k: v for k, v in sdr.items() if not k.startswith("upsampler.")})
    a, b = rnd((1, 16, 24, 24), 7), rnd((1, 16, 24, 24), 8)
    with torch.no_grad():
        y = ref.to(DEV).eval()(a.to(DEV), b.to(DEV))
    xx = torch.cat([a, b], 1)
    for i in range(3):
        xx = O.basic_block(sdr, f"fuse.{i}.", xx, False)
    want = torch.nn.functional.conv2d(xx, sdr["conv_last.weight"], sdr["conv_last.bias"], 1, 1)
    assert O.rel_l2(y.cpu(), want) <= TOL
    hup = HRupsample(16, 3, 4)
    hup.load_state_dict({k: v for k, v in sdr.items() if k.startswith(("upsampler.", "conv_last."))})
    a = rnd((1, 16, 10, 10), 9)
    with torch.no_grad():
        y = hup.to(DEV).eval()(a.to(DEV))
    want = torch.nn.functional.conv2d(O.upsampler(sdr, "upsampler.", a, 4), sdr["conv_last.weight"], sdr["conv_last.bias"], 1, 1)
    assert O.rel_l2(y.cpu(), want) <= TOL


def test_aggregate_kernel(golden_dir):
    from srbh_amd.aggregate import aggregate_torch
    g = gold(golden_dir, "g8_aggregate")
    lab = g["label"].float()
    out = aggregate_torch(lab.to(DEV), 0.25)
    assert out.shape == (64, 64)
    assert torch.allclose(out.cpu(), g["out"], rtol=1e-6, atol=1e-6)
    batch = torch.cat([lab, lab.flip(-1)], 0)
    out2 = aggregate_torch(batch.to(DEV), 0.25)
    assert out2.shape == (2, 64, 64) and torch.allclose(out2[0].cpu(), g["out"], rtol=1e-6, atol=1e-6)


def _train_check(g, tag, module, inputs, tol_grad=5e-5):
    """forward in train mode with autograd, backward of sum(y*w): outputs, input grads, parameter grads and the
    updated running statistics must match what the reference produced (tools/make_golden.py::_train_eval)."""
    module = module.to(DEV).train()
    ins = [t.clone().to(DEV).requires_grad_(True) for t in inputs]
    y = module(*ins)
    assert O.rel_l2(y.detach().cpu(), g[tag + "_train"]) <= TOL
    wgt = rnd(tuple(y.shape), 777).to(DEV)
    (y * wgt).sum().backward()
    for i, t in enumerate(ins):
        e = O.rel_l2(t.grad.cpu(), g[f"{tag}_dx{i}"])
        assert e <= tol_grad, (f"dx{i}", e)
    for k, p in module.named_parameters():
        e = O.rel_l2(p.grad.cpu(), g[f"{tag}_grad_{k}"])
        assert e <= tol_grad, (k, e)
    msd = module.state_dict()
    for k, v in g.items():
        if k.startswith(tag + "_stat_"):
            name = k[len(tag + "_stat_"):]
            assert torch.allclose(msd[name].cpu().double(), v.double(), rtol=1e-5, atol=1e-6), name


@pytest.mark.parametrize("inp,planes,seed,tag,xseed", [(32, 16, 16, "g6_bb32_16", 106), (16, 16, 17, "g6_bb16_16", 107)])
def test_basicblock_backward(inp, planes, seed, tag, xseed, golden_dir):
    from srbh_amd.hrfuse import BasicBlock
    g = gold(golden_dir, "g6_basicblock")
    sd = {}
    synth.basicblock_state_dict(sd, "", inp, planes, seed, "stress")
    blk = BasicBlock(inp, planes)
    blk.load_state_dict(sd, strict=True)
    _train_check(g, tag, blk, [rnd((2, inp, 12, 12), xseed)])


def test_head_backward(golden_dir):
    from srbh_amd.hrfuse import HRfeature, HRfuse_residual
    g = gold(golden_dir, "g7_head")
    m = HRfeature(64, 16, 16)
    m.load_state_dict(synth.hrfeature_state_dict(64, 16, 16, seed=18, mode="stress"), strict=True)
    _train_check(g, "g7_hrfeat", m, [rnd((2, 64, 16, 16), 108)])
    for oc in (1, 7):
        m = HRfuse_residual(16, 16, 16, oc, 4)
        m.load_state_dict(synth.hrfuse_residual_state_dict(16, 16, 16, oc, 4, seed=19 + oc, mode="stress"), strict=True)
        _train_check(g, f"g7_fuse{oc}", m, [rnd((2, 16, 4, 4), 109), rnd((2, 16, 16, 16), 110)])


def test_backward_larger_ragged_vs_oracle_autograd():
    """several tiles per image + ragged edges: compare with torch autograd over the CPU oracle."""
    from srbh_amd.hrfuse import HRfuse_residual
    sd = synth.hrfuse_residual_state_dict(16, 16, 16, 7, 4, seed=31, mode="stress")
    m = HRfuse_residual(16, 16, 16, 7, 4)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    a, b = rnd((2, 16, 19, 23), 1), rnd((2, 16, 76, 92), 2)
    ins = [a.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)]
    y = m(*ins)
    wgt = rnd(tuple(y.shape), 3)
    (y * wgt.to(DEV)).sum().backward()
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    oa, ob = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = O.hrfuse_residual(osd, "", oa, ob, True)
    (yo * wgt).sum().backward()
    assert O.rel_l2(y.detach().cpu(), yo.detach()) <= TOL
    assert O.rel_l2(ins[0].grad.cpu(), oa.grad) <= 1e-4 and O.rel_l2(ins[1].grad.cpu(), ob.grad) <= 1e-4
    for k, p in m.named_parameters():
        assert O.rel_l2(p.grad.cpu(), osd[k].grad) <= 1e-4, k
