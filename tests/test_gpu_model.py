"""GPU: the whole height model (stock-op encoder/decoders + libsrbh head) and one training step of the harness."""
import copy

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


def make_model(seed=0, isaggre=True):
    from srbh_amd.models import SRRegress_Cls_feature
    torch.manual_seed(seed)
    m = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=isaggre,
                              chans_build=7)
    # non-trivial BN statistics everywhere so eval mode is a real test
    g = torch.Generator()
    g.manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_((torch.rand(mod.num_features, generator=g) - 0.5) * 0.2)
            mod.running_var.copy_(0.5 + torch.rand(mod.num_features, generator=g))
    return m


def cpu_reference(m_cpu, x, fea, training):
    """stock-op encoder/decoders on CPU + the oracle's functional head on the same state_dict (autograd-capable)."""
    sd = {k: v for k, v in m_cpu.state_dict(keep_vars=True).items()}
    feats = m_cpu.encoder(x)
    sup = O.hrfeature(sd, "hrfeat.", fea, training)
    hfea = m_cpu.decoder1(*feats)
    aggre = torch.nn.functional.conv2d(hfea, sd["aggre_height.weight"], sd["aggre_height.bias"], 1, 1)
    height = O.hrfuse_residual(sd, "reg.", hfea, sup, training)
    build = O.hrfuse_residual(sd, "seg.", m_cpu.decoder2(*feats), sup, training)
    return height, build, aggre


def test_model_eval_forward_matches_cpu():
    m = make_model().eval()
    x = synth.tiles(2, 8, 64, seed=3)
    fea = torch.randn(2, 64, 256, 256, generator=torch.Generator().manual_seed(4)) * 0.5
    with torch.no_grad():
        want = cpu_reference(copy.deepcopy(m), x, fea, False)
        got = m.to(DEV)(x.to(DEV), fea.to(DEV))
    assert got[0].shape == (2, 1, 256, 256) and got[1].shape == (2, 7, 256, 256) and got[2].shape == (2, 1, 64, 64)
    for a, b, name in zip(got, want, ("height", "build", "aggre")):
        assert O.rel_l2(a.cpu(), b) <= 2e-4, name
    p = make_model(isaggre=False).eval().to(DEV)
    with torch.no_grad():
        out = p(x.to(DEV), fea.to(DEV))
        assert len(out) == 2
        assert p.forward_unsup(x.to(DEV), fea.to(DEV)).shape == (2, 256, 256)
        assert p.forward_nobuild(x.to(DEV), fea.to(DEV)).shape == (2, 1, 256, 256)


@pytest.mark.parametrize("isaggre", [False, True])
@pytest.mark.parametrize("training", [False, True])
def test_forward_unsup_and_nobuild_values_match_cpu(monkeypatch, isaggre, training):
    """a16 (reference mymodels.py:295-337): `forward_unsup` = squeezed height only; `forward_nobuild` = height (+ the
    aggregated height when isaggre) without decoder2 / seg -- VALUES against the CPU reference (stock-op encoder /
    decoder1 on CPU + the oracle head on the same state_dict), eval and train mode."""
    from srbh_amd import encoders
    monkeypatch.setattr(encoders, "DROP_CONNECT", 0.0)          # the only RNG in the model
    x = synth.tiles(2, 8, 64, seed=23)
    fea = torch.randn(2, 64, 256, 256, generator=torch.Generator().manual_seed(24)) * 0.5

    def fresh():
        m = make_model(seed=25, isaggre=isaggre)
        return m.train(training)

    with torch.no_grad():
        mc = fresh()
        sd = dict(mc.state_dict(keep_vars=True))
        feats = mc.encoder(x)
        sup = O.hrfeature(sd, "hrfeat.", fea, training)
        hfea = mc.decoder1(*feats)
        want_h = O.hrfuse_residual(sd, "reg.", hfea, sup, training)
        want_a = torch.nn.functional.conv2d(hfea, sd["aggre_height.weight"], sd["aggre_height.bias"], 1, 1) if isaggre else None
        got_u = fresh().to(DEV).forward_unsup(x.to(DEV), fea.to(DEV))
        got_n = fresh().to(DEV).forward_nobuild(x.to(DEV), fea.to(DEV))
    tol = 2e-4 if not training else 2e-3        # train-mode BN statistics over 2 tiles amplify MIOpen-vs-oneDNN noise
    assert got_u.shape == (2, 256, 256)
    assert O.rel_l2(got_u.cpu(), want_h.squeeze()) <= tol
    if isaggre:
        assert isinstance(got_n, tuple) and len(got_n) == 2
        assert got_n[0].shape == (2, 1, 256, 256) and got_n[1].shape == (2, 1, 64, 64)
        assert O.rel_l2(got_n[0].cpu(), want_h) <= tol and O.rel_l2(got_n[1].cpu(), want_a) <= tol
    else:
        assert torch.is_tensor(got_n) and got_n.shape == (2, 1, 256, 256)
        assert O.rel_l2(got_n.cpu(), want_h) <= tol


def test_model_train_step_gradients_match_cpu_autograd(monkeypatch):
    from srbh_amd import encoders
    monkeypatch.setattr(encoders, "DROP_CONNECT", 0.0)          # the only RNG in the model
    m = make_model(seed=5).train()
    mc = copy.deepcopy(m)
    x = synth.tiles(2, 8, 64, seed=6)
    fea = torch.randn(2, 64, 256, 256, generator=torch.Generator().manual_seed(7)) * 0.5
    w = [torch.randn(2, 1, 256, 256), torch.randn(2, 7, 256, 256), torch.randn(2, 1, 64, 64)]
    outs = cpu_reference(mc, x, fea, True)
    sum((o * ww).sum() for o, ww in zip(outs, w)).backward()
    m = m.to(DEV)
    got = m(x.to(DEV), fea.to(DEV))
    sum((o * ww.to(DEV)).sum() for o, ww in zip(got, w)).backward()
    for a, b in zip(got, outs):
        assert O.rel_l2(a.detach().cpu(), b.detach()) <= 2e-4
    cpu_grads = dict(mc.named_parameters())
    checked = 0
    gmax = max(float(v.grad.norm()) for v in cpu_grads.values() if v.grad is not None)
    for k, p in m.named_parameters():
        if p.grad is None:
            assert cpu_grads[k].grad is None, k
            continue
        e = O.rel_l2(p.grad.cpu(), cpu_grads[k].grad)
        # libsrbh head: fp32 kernels, tight.  Stock-op decoders/encoder: MIOpen-vs-oneDNN noise is amplified through
        # 32 train-mode BatchNorms evaluated on 2x2x2 samples, so only a sanity bound applies there.
        if k.split(".")[0] in ("hrfeat", "reg", "seg", "aggre_height"):
            # (the strict 5e-5 head-gradient checks live in test_gpu_head.py; here the head's inputs come from the
            #  stock-op decoders whose GPU/CPU outputs already differ at the 1e-4 level through train-mode BN)
            assert e <= 2e-2, (k, e)
        else:
            assert bool(torch.isfinite(p.grad).all()), k
            # parameters whose true gradient is ~0 (e.g. a BN bias feeding the next train-mode BN) carry only noise
            if float(cpu_grads[k].grad.norm()) > 1e-3 * gmax:
                assert e <= 5e-2, (k, e)
        checked += 1
    assert checked > 500
    assert m.encoder._conv_head.weight.grad is None               # unused parameter, as upstream
    # BN running statistics moved identically
    for (k, a), (_, b) in zip(m.named_buffers(), mc.named_buffers()):
        if "running" in k and k.split(".")[0] in ("hrfeat", "reg", "seg"):
            assert torch.allclose(a.cpu(), b, rtol=1e-4, atol=1e-5), k


def test_train_harness_step_reduces_loss():
    from srbh_amd.harness import TrainStep, synthetic_batch
    from srbh_amd.rrdbnet import RRDBNet
    net_hr = RRDBNet(3, 3, num_block=2)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=2, seed=2, mode="stress"))
    ts = TrainStep(net_hr.to(DEV), make_model(seed=9).to(DEV), DEV)
    batch = synthetic_batch(4, 11, DEV)
    assert batch[0].shape == (4, 8, 64, 64) and batch[1].shape == (4, 256, 256) and batch[2].shape == (4, 64, 64)
    assert batch[3].dtype == torch.long and int(batch[3].max()) <= 6
    losses = [float(ts(batch)[0]) for _ in range(6)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses


def test_predict_path_sharded_mosaic():
    """config 5 shape: grid cells -> both nets -> device mosaic; two 'ranks' merged == one rank (integer sums)."""
    from srbh_amd.harness import predict_tiles
    from srbh_amd.mosaic import Mosaic
    from srbh_amd.rrdbnet import RRDBNet
    net_hr = RRDBNet(3, 3, num_block=1)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=1, seed=4, mode="stress"))
    net_hr = net_hr.to(DEV).eval()
    model = make_model(seed=12, isaggre=False).to(DEV).eval()
    tiles = synth.tiles(6, 8, 64, seed=13, kind="grid")
    pos = [[0, 0, 64, 64], [64, 0, 64, 64], [128, 0, 32, 64], [0, 64, 64, 16], [64, 64, 64, 16], [32, 32, 64, 48]]
    H, W = 80 * 4, 160 * 4
    one = Mosaic(H, W, 7, DEV)
    assert predict_tiles(net_hr, model, tiles, pos, one, batch=6) == 6
    a, b = Mosaic(H, W, 7, DEV), Mosaic(H, W, 7, DEV)
    predict_tiles(net_hr, model, tiles, pos, a, batch=2, rank=0, world=2)
    predict_tiles(net_hr, model, tiles, pos, b, batch=2, rank=1, world=2)
    a.merge_(b)
    assert torch.equal(a.res_weight, one.res_weight)
    # heights are quantised model outputs: batch-size dependent stock-op algorithms may move a value across a rounding
    # boundary, so compare the final rasters with a 1-LSB allowance on a vanishing fraction
    h1, c1 = one.finalize()
    h2, c2 = a.finalize()
    dh = (h1.int() - h2.int()).abs()
    assert int(dh.max()) <= 1 and float((dh > 0).float().mean()) < 1e-3
    assert float((c1 != c2).float().mean()) < 1e-3
    assert int(one.res_weight.max()) == 2          # the overlapping window really overlapped


def test_predict_graph_replay_equals_eager_launches(monkeypatch):
    """predict_tiles replays one captured HIP graph per full batch (harness._PredictGraph); same kernels, same order: the integer
    mosaic must be identical to the eagerly launched path, the ragged tail goes through the eager path, and a parameter
    update (fused optimizer: no version bump) must trigger a re-capture instead of replaying stale weights."""
    from srbh_amd import harness
    from srbh_amd.mosaic import Mosaic
    from srbh_amd.rrdbnet import RRDBNet
    net_hr = RRDBNet(3, 3, num_block=1)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=1, seed=4, mode="stress"))
    net_hr = net_hr.to(DEV).eval()
    model = make_model(seed=12, isaggre=False).to(DEV).eval()
    tiles = synth.tiles(10, 8, 64, seed=23, kind="grid").to(DEV)
    pos = [[(i % 4) * 48, (i // 4) * 48, 64, 64] for i in range(10)]
    Hh, Ww = 4 * (48 * 2 + 64), 4 * (48 * 3 + 64)

    def run(graph):
        monkeypatch.setattr(harness, "PREDICT_GRAPH", graph)
        m = Mosaic(Hh, Ww, 7, DEV)
        assert harness.predict_tiles(net_hr, model, tiles, pos, m, batch=4) == 10      # 2 full batches + a tail of 2
        return m

    def same(a, b):
        """the forward is not bit-reproducible from run to run (atomics in the pooled reductions: 5e-5 between two eager runs),
        so, as in test_predict_path_sharded_mosaic: final rasters equal up to 1 LSB on a vanishing fraction of the pixels"""
        (h1, c1), (h2, c2) = a.finalize(), b.finalize()
        dh = (h1.int() - h2.int()).abs()
        return (torch.equal(a.res_weight, b.res_weight) and int(dh.max()) <= 1 and float((dh > 0).float().mean()) < 1e-3
                and float((c1 != c2).float().mean()) < 1e-3)

    eager, replay = run(False), run(True)
    pg = model.__dict__["_srbh_predict_graph"]
    assert pg is not None and same(eager, replay)
    again = run(True)
    assert model.__dict__["_srbh_predict_graph"] is pg and same(again, replay)     # cached graph reused
    opt = torch.optim.SGD(model.reg.parameters(), lr=0.5)
    for p in model.reg.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    moved = run(True)
    assert model.__dict__["_srbh_predict_graph"] is not pg
    assert same(moved, run(False)) and not same(moved, replay)
