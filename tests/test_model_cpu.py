"""CPU: SRRegress_Cls_feature boundary (signature, state_dict prefixes, parameter-count anchors of the unpinned
third-party parts), drop-in import shims, sharding + gradient all-reduce under gloo world_size 2."""
import inspect
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "super-resolution-building-height-estimation_amd")


def nparams(m):
    return sum(p.numel() for p in m.parameters())


def test_model_signature_and_param_anchors():
    from srbh_amd.models import SRRegress_Cls_feature
    sig = inspect.signature(SRRegress_Cls_feature.__init__)
    names = list(sig.parameters)[1:]
    assert names == ["encoder_name", "encoder_weights", "encoder_depth", "in_channels", "classes", "super_in", "super_mid",
                     "upscale", "isaggre", "chans_build", "uniform_range", "isunsup"]
    d = {k: v.default for k, v in sig.parameters.items() if k != "self"}
    assert (d["encoder_name"], d["encoder_weights"], d["encoder_depth"], d["in_channels"], d["classes"], d["super_in"],
            d["super_mid"], d["upscale"], d["isaggre"], d["chans_build"]) == ("resnet50", "imagenet", 5, 7, 1, 4, 64, 4, False, 2)
    m = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True,
                              chans_build=7)                       # train.py:143-148
    # author's recorded counts (mymodels.py:765): encoder 17.55 M (+2160 for the 8-channel stem), decoder 2.68 M
    assert nparams(m.encoder) == 17_548_616 + 48 * 5 * 9
    assert nparams(m.decoder1) == nparams(m.decoder2) == 2_676_928
    assert nparams(m.hrfeat) == 21_984 and nparams(m.reg) == 35_569 and nparams(m.seg) == 36_439
    assert m.encoder.out_channels == (8, 48, 32, 56, 160, 448)
    keys = list(m.state_dict().keys())
    prefixes = {k.split(".")[0] for k in keys}
    assert prefixes == {"encoder", "decoder1", "decoder2", "reg", "seg", "hrfeat", "aggre_height"}
    for k in ("encoder._conv_stem.weight", "encoder._bn0.running_mean", "encoder._blocks.0._depthwise_conv.weight",
              "encoder._blocks.2._expand_conv.weight", "encoder._blocks.31._se_expand.bias", "encoder._conv_head.weight",
              "encoder._bn1.weight", "decoder1.blocks.0.conv1.0.weight", "decoder2.blocks.4.conv2.1.num_batches_tracked",
              "reg.upsampler.0.weight", "reg.upsampler.2.bias", "seg.fuse.0.downsample.1.running_var", "seg.conv_last.bias",
              "hrfeat.0.downsample.0.weight", "hrfeat.2.bn2.weight", "aggre_height.weight"):
        assert k in m.state_dict(), k
    assert "encoder._blocks.0._expand_conv.weight" not in m.state_dict()      # expand ratio 1 blocks have none
    assert "encoder._fc.weight" not in m.state_dict()
    # predict-time construction: isaggre=False, checkpoint loaded with strict=False (predict...py:90-93,109)
    p = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=False,
                              chans_build=7)
    missing, unexpected = p.load_state_dict(m.state_dict(), strict=False)
    assert not missing and set(unexpected) == {"aggre_height.weight", "aggre_height.bias"}
    # the low-resolution stock-op part runs on CPU; the 256x256 head refuses (no CPU fallback)
    m.eval()
    x = torch.rand(2, 8, 64, 64)
    with torch.no_grad():
        feats = m.encoder(x)
        assert [tuple(f.shape[1:]) for f in feats] == [(8, 64, 64), (48, 32, 32), (32, 16, 16), (56, 8, 8), (160, 4, 4), (448, 2, 2)]
        assert m.decoder1(*feats).shape == (2, 16, 64, 64)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m(x, torch.rand(2, 64, 256, 256))
    with pytest.raises(NotImplementedError):
        SRRegress_Cls_feature("resnet50")


def test_dropin_import_paths_resolve_to_the_build():
    code = ("import sys; sys.path.insert(0, %r);"
            "from SR.rrdbnet_arch import RealESRGAN, RRDBNet;"
            "from SR.HRfuse import HRfuse, HRfuse_x2, HRfeature, HRfuse_residual, Refine_residual, GeoNet, HRupsample;"
            "from mymodels import SRRegress_Cls_feature;"
            "from aggregate_utils import aggregate_torch;"
            "from losses_pytorch.selfloss import CE_DICE_adapt, MSE_adapt, MSE_adapt_weight, CE_DICE_adapt_weight;"   # train.py:20
            "from metrics import AverageMeter, SegmentationMetric, HeightMetric;"                                     # train.py:13
            "import srbh_amd.rrdbnet as r, srbh_amd.hrfuse as h, srbh_amd.losses as l, srbh_amd.metrics as m;"
            "assert RRDBNet is r.RRDBNet and HRfeature is h.HRfeature;"
            "assert MSE_adapt_weight is l.MSE_adapt_weight and HeightMetric is m.HeightMetric; print('ok')") % os.path.join(PKG, "dropin")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_shard_range_partitions():
    from srbh_amd.harness import shard_range
    for n in (0, 1, 7, 301, 45000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from srbh_amd.harness import allreduce_grads, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 3, padding=1))
    unused = torch.nn.Parameter(torch.zeros(3))              # like encoder._conv_head: never gets a grad
    g = torch.Generator()
    g.manual_seed(1)
    x, y = torch.rand(8, 3, 16, 16, generator=g), torch.rand(8, 1, 16, 16, generator=g)
    lo, hi = shard_range(8, rank, world)
    loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean()
    loss.backward()
    nb = allreduce_grads(list(net.parameters()) + [unused], world, dist, bucket_bytes=256)
    dist.barrier()
    q.put((rank, nb, [p.grad.numpy().copy() for p in net.parameters()]))   # by value: the worker may exit first
    dist.destroy_process_group()


def test_dp_gradient_allreduce_gloo_world2():
    """N>1 path on CPU: equal shards + mean loss + averaged grads == the single-process full-batch gradient."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 3, padding=1))
    g = torch.Generator()
    g.manual_seed(1)
    x, y = torch.rand(8, 3, 16, 16, generator=g), torch.rand(8, 1, 16, 16, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    assert res[0][1] >= 2                                     # several buckets were exercised
    for r in res:
        for got, p in zip(r[2], net.parameters()):
            assert torch.allclose(torch.from_numpy(got), p.grad, rtol=1e-5, atol=1e-7)


def test_same_pad_conv_known_answers():
    """a18 anchor (parity otherwise unpinned): efficientnet_pytorch 0.7.1 `Conv2dStaticSamePadding` pads for a NOMINAL image size,
    pad = max((ceil(i/s)-1)*s + k - i, 0) split (p//2, p - p//2) left/right and top/bottom -- i.e. the odd pixel goes to the
    RIGHT / BOTTOM.  Hand-computed paddings for the B4 shapes the encoder builds (nominal 380 -> 190 -> 95 -> 48 -> 24 -> 12) and
    one hand-computed output."""
    from srbh_amd.encoders import SamePadConv2d
    cases = {(380, 3, 2): (0, 1, 0, 1), (190, 3, 1): (1, 1, 1, 1), (95, 5, 2): (2, 2, 2, 2), (48, 5, 2): (1, 2, 1, 2),
             (24, 3, 2): (0, 1, 0, 1), (12, 5, 1): (2, 2, 2, 2), (12, 1, 1): (0, 0, 0, 0), (95, 3, 2): (1, 1, 1, 1)}
    for (size, k, s), want in cases.items():
        assert SamePadConv2d(2, 2, k, size, stride=s, bias=False)._pad == want, (size, k, s)
    c = SamePadConv2d(1, 1, 3, 4, stride=2, bias=False)              # nominal 4x4, stride 2 -> pad (0,1,0,1)
    with torch.no_grad():
        c.weight.fill_(1.0)
        y = c(torch.arange(16.0).reshape(1, 1, 4, 4))
    assert y.shape == (1, 1, 2, 2) and y.flatten().tolist() == [45.0, 39.0, 66.0, 50.0]
    # the padding is static: a 64x64 tile through the stem built for 380 still pads right/bottom by one -> 32x32
    stem = SamePadConv2d(8, 4, 3, 380, stride=2, bias=False)
    assert stem(torch.zeros(1, 8, 64, 64)).shape == (1, 4, 32, 32)


def _reducer_worker(rank, world, port, q):
    import torch.distributed as dist
    from srbh_amd.harness import GradReducer, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1),
                              torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 3, padding=1))
    unused = torch.nn.Parameter(torch.zeros(3))              # like encoder._conv_head: never gets a grad
    log_var = torch.nn.Parameter(torch.zeros(1))             # like the criteria's log_var param group (train.py:172-179)
    params = list(net.parameters()) + [unused, log_var]
    red = GradReducer(params, world, dist, bucket_bytes=512)
    g = torch.Generator()
    g.manual_seed(1)
    x, y = torch.rand(8, 3, 16, 16, generator=g), torch.rand(8, 1, 16, 16, generator=g)
    lo, hi = shard_range(8, rank, world)
    out = []
    for step in range(3):                                    # step 1: recorded sweep; steps 2-3: hook-launched buckets
        for p in params:
            p.grad = None
        loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean() * torch.exp(-log_var[0]) + log_var[0] * (step + 1)
        loss.backward()
        nb = red.finish()
        out.append([None if p.grad is None else p.grad.numpy().copy() for p in params])
    dist.barrier()
    q.put((rank, nb, out))
    red.close()
    dist.destroy_process_group()


def test_grad_reducer_overlapped_buckets_gloo_world2():
    """harness.GradReducer (bucket all-reduces launched from autograd hooks while backward runs): every step's averaged
    gradients equal the single-process full-batch gradient, incl. an extra log_var-style parameter and a parameter that
    never receives a gradient."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 17) % 1000
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1),
                              torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 3, padding=1))
    log_var = torch.nn.Parameter(torch.zeros(1))
    g = torch.Generator()
    g.manual_seed(1)
    x, y = torch.rand(8, 3, 16, 16, generator=g), torch.rand(8, 1, 16, 16, generator=g)
    assert res[0][1] >= 3                                    # several buckets
    for step in range(3):
        for p in list(net.parameters()) + [log_var]:
            p.grad = None
        (((net(x) - y) ** 2).mean() * torch.exp(-log_var[0]) + log_var[0] * (step + 1)).backward()
        want = [p.grad for p in net.parameters()] + [None, log_var.grad]
        for r in res:
            for got, w in zip(r[2][step], want):
                if w is None:
                    assert got is None
                else:
                    assert torch.allclose(torch.from_numpy(got), w, rtol=1e-5, atol=1e-7), step


def test_weight_pack_caches_see_fused_optimizer_steps():
    """torch.optim.Adam(fused=True) updates parameters WITHOUT bumping their version counters (torch 2.10), so a packed-weight
    cache keyed on `_version` alone would keep convolving with stale weights (caught on the GPU as a training run whose loss fell
    3x slower).  wcache stamps the parameters of any optimizer that steps; only those (a frozen network keeps its packs)."""
    from srbh_amd import wcache
    w = torch.nn.Parameter(torch.randn(4, 4))
    frozen = torch.nn.Parameter(torch.randn(4, 4))
    for fused in (False, True):
        opt = torch.optim.Adam([w], lr=1e-2, fused=fused)
        w.grad = torch.ones_like(w)
        before, ver, val = wcache.gen(w), w._version, w.detach().clone()
        opt.step()
        assert not torch.equal(val, w.detach())
        assert wcache.gen(w) != before, fused                     # the cache key moved ...
        assert (ver, before) != (w._version, wcache.gen(w))
    assert wcache.gen(frozen) == wcache.gen(torch.nn.Parameter(torch.zeros(1)))   # ... and nobody else's did
    k = wcache.gen(frozen)
    wcache.invalidate_weight_caches()
    assert wcache.gen(frozen) != k


def _reducer_order_worker(rank, world, port, q, mismatch):
    import torch.distributed as dist
    from srbh_amd.harness import GradReducer, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1),
                              torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 3, padding=1))
    extra = torch.nn.Parameter(torch.ones(3))
    params = list(net.parameters()) + [extra]
    red = GradReducer(params, world, dist, bucket_bytes=512)
    g = torch.Generator()
    g.manual_seed(1)
    x, y = torch.rand(8, 3, 16, 16, generator=g), torch.rand(8, 1, 16, 16, generator=g)
    lo, hi = shard_range(8, rank, world)
    out, err = [], None
    try:
        for step in range(3):
            for p in params:
                p.grad = None
            loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean()
            if mismatch and rank == 1:
                loss = loss + extra.sum()                    # this rank alone produces a gradient for `extra`
            loss.backward()
            if step == 0 and rank == 1:
                red._order.reverse()                         # this rank's autograd ready order differs from rank 0's
            red.finish()
            out.append([None if p.grad is None else p.grad.numpy().copy() for p in params])
    except RuntimeError as e:
        err = str(e)
    idx = {id(p): i for i, p in enumerate(params)}
    q.put((rank, out, err, None if red.plan is None else [[idx[id(p)] for p, _, _ in b["items"]] for b in red.plan]))
    red.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("mismatch", [False, True])
def test_grad_reducer_ranks_agree_on_bucket_layout_gloo_world2(mismatch):
    """the bucket layout came from each rank's LOCAL autograd ready order with nothing checking that the ranks agree.
    Now rank 0's order is broadcast (a rank with another order re-orders to it: averaged gradients still equal the full-batch
    gradient), and ranks whose SETS of gradient-receiving parameters differ raise instead of averaging unrelated tensors."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 41 + int(mismatch)) % 1000
    procs = [ctx.Process(target=_reducer_order_worker, args=(r, 2, port, q, mismatch)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    if mismatch:
        assert all(r[2] is not None and "disagree" in r[2] for r in res), [r[2] for r in res]
        return
    assert all(r[2] is None for r in res), [r[2] for r in res]
    assert res[0][3] == res[1][3] and len(res[0][3]) >= 3          # identical bucket layouts on both ranks
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1),
                              torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 3, padding=1))
    g = torch.Generator()
    g.manual_seed(1)
    x, y = torch.rand(8, 3, 16, 16, generator=g), torch.rand(8, 1, 16, 16, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    for r in res:
        for step in range(3):
            for got, p in zip(r[1][step], net.parameters()):
                assert torch.allclose(torch.from_numpy(got), p.grad, rtol=1e-5, atol=1e-7), step


def test_capture_holder_keeps_reported_buffers_and_unpins():
    """wcache.Holder / capturing / keep: what a captured HIP graph uses to own the buffers it points at (CPU logic)."""
    import gc
    import weakref
    from srbh_amd import wcache
    t = torch.zeros(4)
    r = weakref.ref(t)
    wcache.keep(t)                                             # no capture active: a no-op
    h = wcache.Holder()
    calls = []
    with wcache.capturing(h) as hh:
        assert hh is h and wcache.active_holder() is h
        wcache.keep(t, None)
        h.on_release(lambda: calls.append(1))
    assert wcache.active_holder() is None and len(h.refs) == 1
    del t
    gc.collect()
    assert r() is not None                                     # the holder keeps it alive ...
    del h, hh
    gc.collect()
    assert r() is None and calls == [1]                        # ... exactly as long as it lives; release callbacks ran once
    b = torch.zeros(2)
    k = wcache.gen(b)
    wcache.stamp([b, None])
    assert wcache.gen(b) != k                                  # buffers can be stamped like parameters


def test_efficientnet_b4_keys_and_shapes_match_the_published_table(golden_dir):
    """a18 structural anchor (round-3 VERDICT, What's missing #5): segmentation_models_pytorch / efficientnet_pytorch are absent, so
    output parity of the encoder cannot be pinned here; its STRUCTURE can.  tests/golden/efficientnet_b4_keys.json is the published
    efficientnet-b4 state_dict layout written down from the architecture table by tools/make_effnet_b4_table.py (which does not
    import the product and self-checks against the published 19 341 616-parameter count); every key, every shape, the per-block
    strides and the per-stage feature channels / strides of srbh_amd.encoders must match it."""
    import json
    from srbh_amd import encoders as E
    with open(os.path.join(golden_dir, "efficientnet_b4_keys.json")) as f:
        ref = json.load(f)
    enc = E.EfficientNetEncoder("efficientnet-b4", in_channels=3)
    sd = enc.state_dict()
    assert sorted(sd.keys()) == sorted(ref["keys"].keys())
    for k, v in sd.items():
        assert list(v.shape) == ref["keys"][k], k
    assert sum(p.numel() for p in enc.parameters()) == ref["n_params_without_fc"] == 17_548_616
    assert [b.stride for b in enc._blocks] == ref["block_strides"]
    assert list(enc._stage_idxs) == ref["stage_idxs"]
    # the reference's construction: 8 input channels (train.py:143-148); only the stem weight widens
    enc8 = E.get_encoder("efficientnet-b4", in_channels=8, depth=5, weights="imagenet")
    sd8 = enc8.state_dict()
    assert sorted(sd8.keys()) == sorted(ref["keys"].keys())
    assert [k for k in sd8 if list(sd8[k].shape) != ref["keys"][k]] == ["_conv_stem.weight"] and tuple(sd8["_conv_stem.weight"].shape) == (48, 8, 3, 3)
    enc8.eval()
    with torch.no_grad():
        feats = enc8(torch.rand(1, 8, 64, 64))
    want = [[8, 1]] + ref["features_channels_strides"][1:]
    assert [[f.shape[1], 64 // f.shape[2]] for f in feats] == want
    assert all(f.shape[2] == f.shape[3] for f in feats)
