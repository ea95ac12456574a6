"""CPU: the oracle (oracle/srbh_oracle.py) replays the fixtures captured from the imported reference
(tools/make_golden.py).  Weights are regenerated from seeds (oracle/synth.py), inputs likewise."""
import os

import numpy as np
import pytest
import torch

from oracle import srbh_oracle as O
from oracle import synth

TOL = 2e-6  # fp32 summation-order noise between machines / thread counts


def rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def test_g1_g2_rdb_rrdb(golden_dir):
    g = load(golden_dir, "g1_g2_rdb")
    sd = synth.rrdbnet_state_dict(num_block=1, seed=11, mode="stress")
    assert O.rel_l2(O.rdb(sd, "body.0.rdb1.", rand((1, 64, 16, 16), 101)), g["rdb_out"]) <= TOL
    assert O.rel_l2(O.rrdb(sd, "body.0.", rand((2, 64, 12, 12), 102)), g["rrdb_out"]) <= TOL


@pytest.mark.parametrize("scale,hw", [(4, 8), (2, 16), (1, 16)])
def test_g3_small_net(golden_dir, scale, hw):
    g = load(golden_dir, "g3_rrdbnet_small")
    sd = synth.rrdbnet_state_dict(num_block=2, scale=scale, seed=12, mode="stress")
    x = rand((1, 3, hw, hw), 103 + scale, 0.0, 1.0)
    assert O.rel_l2(O.rrdbnet_forward_feature(sd, x, scale), g[f"ff_s{scale}"]) <= TOL
    assert O.rel_l2(O.rrdbnet_forward(sd, x, scale), g[f"fw_s{scale}"]) <= TOL


def full_net_summary(y):
    out = {"ch_mean": y.double().mean((0, 2, 3)), "ch_std": y.double().std((0, 2, 3)),
           "row_sum": y[0].double().sum((0, 2))}
    for name, (r, c) in {"tl": (0, 0), "tr": (0, 248), "bl": (248, 0), "br": (248, 248), "ce": (124, 124)}.items():
        out["crop_" + name] = y[0, :, r:r + 8, c:c + 8]
    return out


@pytest.mark.parametrize("mode", ["init", "stress"])
def test_g4_full_net(golden_dir, mode):
    g = load(golden_dir, f"g4_rrdbnet_full_{mode}")
    sd = synth.rrdbnet_state_dict(seed=1337, mode=mode)
    assert len(sd) == 702 and sum(v.numel() for v in sd.values()) == 16_697_987
    x = synth.tiles(1, 8, 64, seed=1337)[:, :3]
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    y = O.rrdbnet_forward_feature(sd, x)
    for k, v in full_net_summary(y).items():
        assert O.rel_l2(v, g[k]) <= 5e-6, k
    assert abs(float(y.double().sum()) - float(g["checksum"])) <= 1e-5 * float(y.double().abs().sum())


def test_g5_index_maps_bit_exact(golden_dir):
    g = load(golden_dir, "g5_index_maps")
    x = torch.arange(2 * 64 * 5 * 7, dtype=torch.float32).reshape(2, 64, 5, 7)
    assert torch.equal(O.pixel_shuffle(x, 2), g["ps2"])
    assert torch.equal(O.nearest2x(x), g["nearest2"])
    xu = torch.arange(1 * 3 * 8 * 12, dtype=torch.float32).reshape(1, 3, 8, 12)
    assert torch.equal(O.pixel_unshuffle(xu, 2), g["unshuffle2"])
    assert torch.equal(O.pixel_unshuffle(xu, 4), g["unshuffle4"])
    with pytest.raises(AssertionError):
        O.pixel_unshuffle(torch.zeros(1, 1, 5, 4), 2)
    sd = synth.hrfuse_residual_state_dict(16, 16, 16, 1, 4, seed=15, mode="stress")
    assert O.rel_l2(O.upsampler(sd, "upsampler.", rand((1, 16, 6, 6), 105), 4), g["upsampler_out"]) <= TOL
    with pytest.raises(NotImplementedError):
        O.upsampler(sd, "upsampler.", x, 3)


def _replay_train_eval(g, tag, sd, inputs, fn):
    ye = fn(synth.clone_sd(sd), False, *inputs)
    assert O.rel_l2(ye, g[tag + "_eval"]) <= TOL
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    ins = [t.clone().requires_grad_(True) for t in inputs]
    yt = fn(osd, True, *ins)
    assert O.rel_l2(yt, g[tag + "_train"]) <= TOL
    (yt * rand(tuple(yt.shape), 777)).sum().backward()
    for i, t in enumerate(ins):
        assert O.rel_l2(t.grad, g[f"{tag}_dx{i}"]) <= 1e-5
    for k, v in osd.items():
        gk = f"{tag}_grad_{k}"
        if gk in g:
            assert O.rel_l2(v.grad, g[gk]) <= 2e-5, k
        sk = f"{tag}_stat_{k}"
        if sk in g:
            assert torch.allclose(v.detach().double(), g[sk].double(), rtol=1e-5, atol=1e-6), k


def test_g6_basicblock(golden_dir):
    g = load(golden_dir, "g6_basicblock")
    sd = {}
    synth.basicblock_state_dict(sd, "", 32, 16, 16, "stress")
    _replay_train_eval(g, "g6_bb32_16", sd, [rand((2, 32, 12, 12), 106)], lambda s, tr, x: O.basic_block(s, "", x, tr))
    sd = {}
    synth.basicblock_state_dict(sd, "", 16, 16, 17, "stress")
    _replay_train_eval(g, "g6_bb16_16", sd, [rand((2, 16, 12, 12), 107)], lambda s, tr, x: O.basic_block(s, "", x, tr))


def test_g7_head(golden_dir):
    g = load(golden_dir, "g7_head")
    sd = synth.hrfeature_state_dict(64, 16, 16, seed=18, mode="stress")
    assert len(sd) == 42
    _replay_train_eval(g, "g7_hrfeat", sd, [rand((2, 64, 16, 16), 108)], lambda s, tr, x: O.hrfeature(s, "", x, tr))
    for oc in (1, 7):
        sd = synth.hrfuse_residual_state_dict(16, 16, 16, oc, 4, seed=19 + oc, mode="stress")
        assert len(sd) == 48
        _replay_train_eval(g, f"g7_fuse{oc}", sd, [rand((2, 16, 4, 4), 109), rand((2, 16, 16, 16), 110)],
                           lambda s, tr, a, b: O.hrfuse_residual(s, "", a, b, tr))


def test_g8_aggregate(golden_dir):
    g = load(golden_dir, "g8_aggregate")
    lab = g["label"].float()
    out = O.aggregate_torch(lab, 0.25)
    assert out.shape == (64, 64)
    assert torch.equal(out, g["out"])
    # for non-negative labels it is a 4x4 mean pool
    pos = lab.clamp_min(0)
    assert torch.allclose(O.aggregate_torch(pos, 0.25), torch.nn.functional.avg_pool2d(pos, 4).squeeze(), atol=1e-4)


def test_mosaic_oracle_matches_reference_predict_outputs(golden_dir):
    """The reference's predict_whole_image_grid itself produced g12_mosaic.npz (see tools/make_golden.py::g_mosaic)."""
    import os
    import numpy as np
    from oracle.mosaic_oracle import MosaicOracle, synthetic_city
    g = np.load(os.path.join(golden_dir, "g12_mosaic.npz"))
    ypred, logits, pos, lr_w, lr_h = synthetic_city()
    o = MosaicOracle(lr_h * 4, lr_w * 4, 7)
    o.add(ypred, logits, pos)
    h, b = o.finalize()
    assert np.array_equal(h, g["height"]) and np.array_equal(b, g["build"])
