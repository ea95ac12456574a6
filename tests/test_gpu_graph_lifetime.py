"""GPU: a captured HIP graph must OWN the device buffers it points at, and caches must notice updates made behind the
version counters (round-2 VERDICT "what's weak" 1-2).

* the multi-city predict sequence of predict_realesanet_feature_globe.py:221-233 -- every city has another ragged tail --
  through harness.predict_tiles with the graph path on: the round-2 code kept RRDBNet workspaces in a 2-entry LRU, the
  second distinct tail size evicted (freed) the workspace baked into the captured graph, and later replays wrote the
  trunk's activations into freed memory.  Here: tails 32/64/96 in the evicting order under a workspace budget small enough
  to force eviction; mosaics must equal the eager run and the trunk's status must be clean.
* TrainStep(graph=True) with foreign geometries of the frozen net between replays; eval after replays sees the trained
  weights (packed-weight caches, folded BatchNorm affines, predict graph); a ragged batch raises instead of resizing.
* RealESRGAN.model_ema: net_g_ema's forward follows the EMA weights.
"""
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nets(num_block=1, isaggre=False, seed=5):
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    net_hr = RRDBNet(3, 3, num_block=num_block)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=num_block, seed=seed, mode="init"))
    torch.manual_seed(9)
    model = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=isaggre,
                                  chans_build=7)
    return net_hr.to(DEV).eval(), model.to(DEV)


def _city(n, seed):
    gw = 12
    g = torch.Generator(device=DEV).manual_seed(seed)
    tiles = torch.randn((n, 8, 64, 64), generator=g, device=DEV) * 0.25 + 0.35
    pos = [[(i % gw) * 48, (i // gw) * 48, 64, 64] for i in range(n)]
    gh = (n + gw - 1) // gw
    return tiles, pos, (gh * 48 + 16) * 4, (gw * 48 + 16) * 4


def _run_cities(net_hr, model, sizes, graph):
    from srbh_amd import harness
    from srbh_amd.mosaic import Mosaic
    harness.PREDICT_GRAPH = graph
    out = []
    try:
        for i, n in enumerate(sizes):
            tiles, pos, H, W = _city(n, 100 + i)
            m = Mosaic(H, W, 7, DEV)
            harness.predict_tiles(net_hr, model, tiles, pos, m, batch=128, pad_to=32)     # (ends with check_status())
            h, b = m.finalize()
            out.append((m.res_weight.clone(), m.res_height.clone(), h.to(torch.int32).clone(), b.clone()))
    finally:
        harness.PREDICT_GRAPH = True
    return out


def test_multi_city_sequence_graph_path_equals_eager(monkeypatch):
    from srbh_amd.rrdbnet import RRDBNet
    net_hr, model = _nets()
    model.eval()
    # 2 GiB (B=128) + 0.5 + 1.0 + 1.5 GiB of tail workspaces against a 3 GiB budget: tails MUST evict one another, and would evict
    # the graph's B=128 workspace if it were not pinned
    monkeypatch.setattr(RRDBNet, "WS_BUDGET_BYTES", 3 * 2 ** 30)
    sizes = [128 + 20, 128 + 50, 128 + 90, 256, 128 + 20, 128 + 120, 2 * 128 + 40, 128 + 70]     # tails 32, 64, 96, -, 32, 128, 64, 96
    got = _run_cities(net_hr, model, sizes, graph=True)
    pg = model.__dict__.get("_srbh_predict_graph")
    assert pg is not None, "the graph path did not run"
    key128 = (128, 64, 64, 0, torch.device(DEV))
    assert key128 in net_hr._workspaces and net_hr.__dict__["_ws_pins"].get(key128), "the captured graph's workspace must stay pinned"
    ws_ptr = net_hr._workspaces[key128].data_ptr()
    assert any(t.data_ptr() == ws_ptr for t in pg.holder.refs if torch.is_tensor(t))
    assert len(net_hr._workspaces) < 4, "the budget was meant to force evictions among the tail shapes"
    net_hr.check_status()
    want = _run_cities(net_hr, model, sizes, graph=False)
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g[0], w[0]), f"city {i}: weight mosaic"
        # the stock-op encoder may pick another algorithm for a padded batch shape than for a captured one: the quantised sums may
        # differ by an LSB per contribution, the finalised height by at most 1 (0.1 m)
        assert int((g[2] - w[2]).abs().max()) <= 1, f"city {i}: height"
        assert float((g[3] != w[3]).float().mean()) <= 1e-3, f"city {i}: class map"
        assert bool((g[1] >= 0).all())
    # dropping the graph un-pins its workspace
    model.__dict__["_srbh_predict_graph"] = None
    del pg
    import gc
    gc.collect()
    assert not net_hr.__dict__["_ws_pins"].get(key128)


def test_predict_graph_split_over_two_streams_equals_the_single_graph(monkeypatch):
    """round 4: harness._PredictGraph replays the encoder / decoders as their own graph on a second stream next to trunk + HRfeature,
    reg / seg after the join (harness.PREDICT_SPLIT).  Same kernels on the same inputs: the mosaics of a three-city sequence equal the
    single-graph replay's bit for bit (isaggre model: the third output crosses the streams too), twice over (buffer reuse across
    batches: the join orders the next batch's staging behind the previous batch's readers)."""
    from srbh_amd import harness
    net_hr, model = _nets(isaggre=True)
    model.eval()
    sizes = [3 * 128, 128 + 40, 2 * 128, 5 * 128 + 7]
    res = {}
    # (round 6) harness.PREDICT_AHEAD: batch k + 1's encoder / decoders launched behind batch k's trunk, two sets of buffers by the parity of
    # the batch number -- 3, 1, 2 and 5 full batches per city: both parities, a city that ends on either, a pass launched ahead for every batch
    # but a city's first
    for split, ahead in ((True, True), (True, False), (False, False), (True, True)):
        monkeypatch.setattr(harness, "PREDICT_SPLIT", split)
        monkeypatch.setattr(harness, "PREDICT_AHEAD", ahead)
        model.__dict__["_srbh_predict_graph"] = None
        got = _run_cities(net_hr, model, sizes, graph=True)
        pg = model.__dict__["_srbh_predict_graph"]
        assert pg is not None and pg.split == split and pg.ahead == (split and ahead) and len(pg.out) == 3
        assert not pg.ahead or (pg.n == 11 and not any(pg.pending))
        res.setdefault((split, ahead), []).append(got)
    model.__dict__["_srbh_predict_graph"] = None
    for a, b in ((res[(True, True)][0], res[(False, False)][0]), (res[(True, False)][0], res[(False, False)][0]), (res[(True, True)][0], res[(True, True)][1])):
        for i, (g, w) in enumerate(zip(a, b)):
            for u, v in zip(g, w):
                assert torch.equal(u, v), f"city {i}"
    net_hr.check_status()


def test_workspace_budget_evicts_lru_but_never_a_pinned_one(monkeypatch):
    from srbh_amd import wcache
    from srbh_amd.rrdbnet import RRDBNet
    net_hr, _ = _nets()
    monkeypatch.setattr(RRDBNet, "WS_BUDGET_BYTES", int(1.2 * 2 ** 30))
    x = torch.rand((64, 3, 64, 64), device=DEV)
    holder = wcache.Holder()
    with torch.no_grad():
        with wcache.capturing(holder):
            net_hr.forward_feature(x[:32])                    # 0.5 GiB, pinned by the holder
        net_hr.forward_feature(x[:64])                        # 1.0 GiB: over budget, the only unpinned candidate is itself -> kept
        net_hr.forward_feature(x[:16])
        keys = [k[0] for k in net_hr._workspaces]
        assert 32 in keys and 16 in keys and 64 not in keys, keys
        holder.release()
        net_hr.forward_feature(x[:64])
        keys = [k[0] for k in net_hr._workspaces]
        assert 64 in keys and 32 not in keys, keys
    net_hr.check_status()


def test_trainstep_graph_survives_foreign_geometries_and_eval_sees_trained_weights():
    from srbh_amd.harness import TrainStep, synthetic_batch
    from srbh_amd import wcache
    net_hr, net = _nets(num_block=1, isaggre=True)
    ts = TrainStep(net_hr, net, DEV, graph=True, status_every=0)
    B = 8
    batch = synthetic_batch(B, 11, DEV)
    probe = synthetic_batch(4, 12, DEV)[0]

    def eval_heights(fresh):
        if fresh:
            wcache.invalidate_weight_caches()
        net.eval()
        with torch.no_grad():
            h = net(probe, net_hr.forward_feature(probe[:, :3]))[0].clone()
        net.train()
        return h

    losses = [float(ts(batch)[0]) for _ in range(5)]          # 3 eager steps, the capture, one replay
    assert ts._graph is not None
    h_a = eval_heights(False)
    x = torch.rand((48, 3, 64, 64), device=DEV)
    with torch.no_grad():                                     # two foreign geometries of the frozen net between replays
        net_hr.forward_feature(x[:16])
        net_hr.forward_feature(x[:48])
    losses += [float(ts(batch)[0]) for _ in range(6)]
    net_hr.check_status()
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    h_b = eval_heights(False)                                 # caches as the replays left them
    h_c = eval_heights(True)                                  # everything repacked from the live parameters / buffers
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())      # noqa: E731
    moved, stale = rel(h_a, h_c), rel(h_b, h_c)
    # (two evals of ONE state agree to ~3e-5 here, not bit for bit -- measured: atomics in the stock-op encoder's kernels, amplified by
    # the fp16-operand head; six Adam steps at lr 1e-3 from a random init move the prediction by ~0.5)
    assert moved >= 1e-2, f"six more optimizer steps must change the prediction (moved {moved:.2e})"
    assert stale <= max(2e-4, 1e-3 * moved), \
        f"eval after graph replays used stale packed weights / BatchNorm affines: {stale:.2e} from a fresh-cache eval (training moved it {moved:.2e})"
    with pytest.raises(ValueError, match="ONE batch geometry"):
        ts(synthetic_batch(B - 2, 13, DEV))
    # constructing / running a TrainStep leaves the process-wide head precision alone
    from srbh_amd import hrfuse
    assert hrfuse._HEAD_PRECISION["mode"] == "auto"


def test_net_g_ema_forward_follows_model_ema():
    from srbh_amd.rrdbnet import RealESRGAN
    gan = RealESRGAN(num_block=1, device=DEV, is_train=True, ema_decay=0.5)
    x = torch.rand((1, 3, 16, 16), device=DEV)
    with torch.no_grad():
        y0 = gan.net_g_ema(x).clone()
        for p in gan.net_g.parameters():
            p.mul_(1.5)
        gan.model_ema(0.5)
        y1 = gan.net_g_ema(x).clone()
        ref = {k: v.clone() for k, v in gan.net_g_ema.state_dict().items()}
        from srbh_amd.rrdbnet import RRDBNet
        fresh = RRDBNet(3, 3, num_block=1).to(DEV)
        fresh.load_state_dict(ref)
        y2 = fresh.eval()(x)
    assert not torch.equal(y0, y1), "net_g_ema kept convolving with the weights packed at its first call"
    assert torch.equal(y1, y2)
