"""GPU parity of RRDBNet.forward_feature / forward (libsrbh) against the CPU oracle and the fixtures
captured from the imported reference.  Tolerance (BASELINE.json north_star): <= 1e-3 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import srbh_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL_TOL = 1e-3   # stated tolerance of the fp16-operand / fp32-accumulate path vs the fp32 reference


def build(sd, **kw):
    from srbh_amd.rrdbnet import RRDBNet
    net = RRDBNet(3, 3, **kw)
    net.load_state_dict(sd, strict=True)
    return net.to(DEV).eval()


def rnd(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


@pytest.mark.parametrize("scale,hw,B", [(4, 8, 1), (2, 16, 1), (1, 16, 1), (4, 64, 3), (4, 20, 2)])
def test_small_net_forward_and_feature(scale, hw, B, golden_dir):
    sd = synth.rrdbnet_state_dict(num_block=2, scale=scale, seed=12, mode="stress")
    net = build(sd, scale=scale, num_block=2)
    x = rnd((B, 3, hw, hw), 103 + scale)
    with torch.no_grad():
        ff = net.forward_feature(x.to(DEV))
        fw = net(x.to(DEV))
    net.check_status()
    s = 4 * hw // {4: 1, 2: 2, 1: 4}[scale]
    assert ff.shape == (B, 64, s, s) and fw.shape == (B, 3, s, s)
    assert O.rel_l2(ff.cpu(), O.rrdbnet_forward_feature(sd, x, scale)) <= REL_TOL
    assert O.rel_l2(fw.cpu(), O.rrdbnet_forward(sd, x, scale)) <= REL_TOL
    if B == 1 and hw in (8, 16) and (scale, hw) in ((4, 8), (2, 16), (1, 16)):
        g = np.load(os.path.join(golden_dir, "g3_rrdbnet_small.npz"))
        assert O.rel_l2(ff.cpu(), torch.from_numpy(g[f"ff_s{scale}"])) <= REL_TOL   # the reference's own output
        assert O.rel_l2(fw.cpu(), torch.from_numpy(g[f"fw_s{scale}"])) <= REL_TOL


@pytest.mark.parametrize("mode", ["init", "stress"])
def test_full_net_config1_against_reference_fixture(mode, golden_dir):
    """BASELINE config 1: one 64x64 tile through the 23-block net; compare with the reference's crops / statistics."""
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, f"g4_rrdbnet_full_{mode}.npz")).items()}
    sd = synth.rrdbnet_state_dict(seed=1337, mode=mode)
    net = build(sd)
    x = synth.tiles(1, 8, 64, seed=1337)[:, :3].contiguous()
    with torch.no_grad():
        y = net.forward_feature(x.to(DEV)).cpu()
    net.check_status()
    assert y.shape == (1, 64, 256, 256)
    for name, (r, c) in {"tl": (0, 0), "tr": (0, 248), "bl": (248, 0), "br": (248, 248), "ce": (124, 124)}.items():
        assert O.rel_l2(y[0, :, r:r + 8, c:c + 8], g["crop_" + name]) <= REL_TOL, name
    assert O.rel_l2(y.double().mean((0, 2, 3)), g["ch_mean"]) <= REL_TOL
    assert O.rel_l2(y.double().std((0, 2, 3)), g["ch_std"]) <= REL_TOL
    assert O.rel_l2(y[0].double().sum((0, 2)), g["row_sum"]) <= REL_TOL
    # and the whole map against the oracle run here
    want = O.rrdbnet_forward_feature(sd, x)
    e = O.rel_l2(y, want)
    print(f"full net [{mode}] rel-L2 {e:.3e} max-rel {O.max_rel(y, want):.3e}")
    assert e <= REL_TOL and O.max_rel(y, want) <= 5e-3


def test_batch_independence_and_determinism():
    """tiles are independent: a tile's output does not depend on its batch neighbours or position."""
    sd = synth.rrdbnet_state_dict(num_block=3, seed=5, mode="stress")
    net = build(sd, num_block=3)
    x = synth.tiles(5, 3, 64, seed=9).to(DEV)
    with torch.no_grad():
        y = net.forward_feature(x)
        y2 = net.forward_feature(x)
        y_single = net.forward_feature(x[3:4])
        y_perm = net.forward_feature(x.flip(0))
    net.check_status()
    assert torch.equal(y, y2)
    assert torch.equal(y[3:4], y_single)
    assert torch.equal(y_perm.flip(0), y)
    assert y.is_contiguous(memory_format=torch.channels_last)


def test_weight_update_is_picked_up():
    sd = synth.rrdbnet_state_dict(num_block=1, seed=6, mode="stress")
    net = build(sd, num_block=1)
    x = rnd((1, 3, 16, 16), 1).to(DEV)
    with torch.no_grad():
        y0 = net.forward_feature(x)
        net.conv_hr.bias.add_(1.0)
        y1 = net.forward_feature(x)
    assert torch.allclose(y1, y0 + 1.0, atol=1e-5)
    sd2 = synth.rrdbnet_state_dict(num_block=1, seed=66, mode="stress")
    net.load_state_dict(sd2)
    with torch.no_grad():
        y2 = net.forward_feature(x)
    assert O.rel_l2(y2.cpu(), O.rrdbnet_forward_feature(sd2, x.cpu())) <= REL_TOL


def test_full_size_batch32_properties():
    """BASELINE config 2 size (B=32): size-independent checks -- finite, equals the B=1 run tile by tile for a sample."""
    sd = synth.rrdbnet_state_dict(seed=1337, mode="init")
    net = build(sd)
    x = synth.tiles(32, 8, 64, seed=1337)[:, :3].contiguous().to(DEV)
    with torch.no_grad():
        y = net.forward_feature(x)
        assert y.shape == (32, 64, 256, 256) and bool(torch.isfinite(y).all())
        for i in (0, 17, 31):
            assert torch.equal(net.forward_feature(x[i:i + 1]), y[i:i + 1])
    net.check_status()


def test_persistent_and_per_layer_paths_agree(monkeypatch):
    """the persistent trunk kernel (default) and the per-layer launch sequence (SRBH_PERSISTENT=0) are the same
    arithmetic in the same order: bit-identical outputs, incl. a batch that needs several persistent sub-launches."""
    sd = synth.rrdbnet_state_dict(num_block=3, seed=21, mode="stress")
    net = build(sd, num_block=3)
    for B, hw in ((3, 64), (40, 64), (2, 40)):
        x = synth.tiles(B, 3, hw, seed=22).to(DEV)
        with torch.no_grad():
            monkeypatch.setenv("SRBH_PERSISTENT", "1")
            y1 = net.forward_feature(x)
            net.check_status()
            monkeypatch.setenv("SRBH_PERSISTENT", "0")
            y0 = net.forward_feature(x)
        assert torch.equal(y0, y1), (B, hw)


def test_strict_fp32_path_matches_reference_and_bounds_the_fast_path(golden_dir):
    """precision='f32': exact-fp32 matrix-core convs -> agrees with the fp32 reference to rounding (1e-5), and is the
    on-device yardstick for the default fp16-operand path (<= 1e-3)."""
    g = np.load(os.path.join(golden_dir, "g3_rrdbnet_small.npz"))
    for scale, hw in ((4, 8), (2, 16)):
        sd = synth.rrdbnet_state_dict(num_block=2, scale=scale, seed=12, mode="stress")
        net = build(sd, scale=scale, num_block=2)
        net.precision = "f32"
        x = rnd((1, 3, hw, hw), 103 + scale)
        with torch.no_grad():
            ff, fw = net.forward_feature(x.to(DEV)), net(x.to(DEV))
        assert O.rel_l2(ff.cpu(), torch.from_numpy(g[f"ff_s{scale}"])) <= 1e-5
        assert O.rel_l2(fw.cpu(), torch.from_numpy(g[f"fw_s{scale}"])) <= 1e-5
    gg = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "g4_rrdbnet_full_stress.npz")).items()}
    sd = synth.rrdbnet_state_dict(seed=1337, mode="stress")
    net = build(sd)
    x = synth.tiles(2, 8, 64, seed=1337)[:, :3].contiguous().to(DEV)
    with torch.no_grad():
        fast = net.forward_feature(x)
        net.precision = "f32"
        strict = net.forward_feature(x)
    assert strict.shape == fast.shape == (2, 64, 256, 256)
    assert O.rel_l2(strict[0, :, 124:132, 124:132].cpu(), gg["crop_ce"]) <= 2e-5      # vs the reference itself
    assert O.rel_l2(strict[:1].double().mean((0, 2, 3)).cpu(), gg["ch_mean"]) <= 2e-5
    e = O.rel_l2(fast.cpu(), strict.cpu())
    print(f"fp16-operand path vs strict fp32 path: rel-L2 {e:.3e}")
    assert e <= REL_TOL
