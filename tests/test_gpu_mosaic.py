"""GPU: quantise + integer mosaic epilogue against the numpy restatement of the reference's lines."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_batch(n, seed, th=256, tw=256):
    g = torch.Generator()
    g.manual_seed(seed)
    y = torch.randn(n, 1, th, tw, generator=g) * 8 + 3            # negatives exercise the clamp
    y[:, :, ::7, ::5] = torch.round(y[:, :, ::7, ::5] * 10) / 10 + 0.05   # values whose x10 lands on .5 (half-even)
    b = torch.randn(n, 7, th, tw, generator=g) * 2
    return y, b


def test_mosaic_matches_reference_numpy_semantics_and_is_shard_invariant():
    from oracle.mosaic_oracle import MosaicOracle
    from srbh_amd.mosaic import Mosaic
    Hc, Wc = 40, 52                                              # city size in LR cells -> mosaics are x4
    H, W = Hc * 4, Wc * 4
    # overlapping 64x64-cell... small: tiles of 16x16 cells (64x64 px), some cropped at the city edge (xcount<16)
    pos = []
    for yo in range(0, Hc, 12):
        for xo in range(0, Wc, 12):
            pos.append([xo, yo, min(16, Wc - xo), min(16, Hc - yo)])
    n = len(pos)
    y, b = make_batch(n, 1, 64, 64)
    ora = MosaicOracle(H, W, 7)
    ora.add(y, b, pos)
    want_h, want_b = ora.finalize()
    full = Mosaic(H, W, 7, DEV)
    full.add(y.to(DEV), b.to(DEV), pos)
    got_h, got_b = full.finalize()
    # integer accumulators: height and weight exactly, class sums within the exp() rounding of softmax (<=1 LSB, rare)
    assert np.array_equal((full.res_height.cpu().numpy() & 0xffff).astype(np.uint16), ora.res_height)
    assert np.array_equal((full.res_weight.cpu().numpy() & 0xff).astype(np.uint8), ora.res_weight)
    db = np.abs((full.res_build.cpu().numpy() & 0xffff).astype(np.int64) - ora.res_build.astype(np.int64))
    assert db.max() <= 2 and (db > 0).mean() < 1e-3
    assert np.array_equal(got_h.cpu().numpy(), want_h)
    assert (got_b.cpu().numpy() != want_b).mean() < 1e-3
    # sharding / order invariance: two shards in reversed order, merged by addition -> bit-identical mosaics
    a, c = Mosaic(H, W, 7, DEV), Mosaic(H, W, 7, DEV)
    half = n // 2
    idx = list(range(n))[::-1]
    ia, ic = idx[:half], idx[half:]
    a.add(y[ia].to(DEV), b[ia].to(DEV), [pos[i] for i in ia])
    c.add(y[ic].to(DEV), b[ic].to(DEV), [pos[i] for i in ic])
    a.merge_(c)
    assert torch.equal(a.res_height, full.res_height) and torch.equal(a.res_build, full.res_build)
    assert torch.equal(a.res_weight, full.res_weight)
    h2, b2 = a.finalize()
    assert torch.equal(h2.view(torch.int16), got_h.view(torch.int16)) and torch.equal(b2, got_b)


def test_mosaic_matches_reference_predict_outputs(golden_dir):
    """g12_mosaic.npz holds what the reference's own predict_whole_image_grid wrote for the synthetic city (tools/
    make_golden.py ran it with fake networks / loader / raster writers): heights bit-exact; the class map may differ
    only where the GPU softmax's exp rounding flips a x255 rounding or an argmax tie."""
    import os
    from oracle.mosaic_oracle import synthetic_city
    from srbh_amd.mosaic import Mosaic
    g = np.load(os.path.join(golden_dir, "g12_mosaic.npz"))
    ypred, logits, pos, lr_w, lr_h = synthetic_city()
    assert np.array_equal(pos.numpy(), g["pos"]) and lr_w == int(g["lr_w"]) and lr_h == int(g["lr_h"])
    m = Mosaic(lr_h * 4, lr_w * 4, 7, DEV)
    m.add(ypred.to(DEV), logits.to(DEV), pos.tolist())
    h, b = m.finalize()
    assert np.array_equal(h.cpu().numpy().astype(np.uint16), g["height"])
    assert (b.cpu().numpy() != g["build"]).mean() < 1e-3
