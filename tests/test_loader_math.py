"""Loader-side tensor math: hierweight known answers on CPU, device kernels on GPU."""
import os

import numpy as np
import pytest
import torch

HIR = (0, 3, 12, 21, 30, 60, 90, 256)


def test_hierweight_known_answers(golden_dir):
    from srbh_amd import loader_math as LM
    g = np.load(os.path.join(golden_dir, "g9_hierweight.npz"))
    stats = g["stats"]
    h255 = (0, 3, 12, 21, 30, 60, 90, 255)
    assert np.allclose(LM.hierweight(stats, h255), g["sqrt255"], rtol=1e-12)
    assert np.allclose(LM.hierweight_simple(stats, h255), g["simple255"], rtol=1e-12)
    assert np.allclose(LM.hierweight(stats, HIR), g["sqrt256"], rtol=1e-12)
    assert np.allclose(LM.hierweight_simple(stats, HIR), g["simple256"], rtol=1e-12)
    # the author's expected vectors (BH_loader.py:1122-1129) and the train.py default hierarchy
    assert np.allclose(g["sqrt255"], [0.08743518, 0.26821995, 0.32067124, 0.73515255, 0.98135007, 1.60267172, 3.0044993], atol=1e-7)
    assert np.allclose(g["sqrt256"], [0.08878965, 0.272375, 0.32563883, 0.74654097, 0.99655239, 1.62749907, 2.94260409], atol=1e-7)
    assert np.array_equal(LM.hierweight_equal(stats, HIR), np.ones(7))
    assert abs(LM.hierweight(stats, HIR).sum() - 7.0) < 1e-9        # the rescale makes the weights sum to the class count


@pytest.mark.gpu
def test_label_prep_and_normalize_kernels(golden_dir):
    from oracle import loader_oracle as LO
    from srbh_amd import loader_math as LM
    g = np.load(os.path.join(golden_dir, "g9_hierweight.npz"))
    hw = LM.hierweight(g["stats"], HIR)
    p = torch.from_numpy(g["stats"] / g["stats"].sum())
    gen = torch.Generator()
    gen.manual_seed(5)
    lab = torch.multinomial(p, 3 * 256 * 256, replacement=True, generator=gen).reshape(3, 256, 256).to(torch.uint8)
    prep = LM.LabelPrep(HIR, hw, "cuda:0")
    hf, ha, build, wt, wa = prep(lab.to("cuda:0"))
    for i in range(3):
        oh, oha, ob, ow, owa = LO.label_prep(lab[i].numpy(), HIR, hw)
        assert torch.equal(hf[i].cpu(), oh) and torch.equal(build[i].cpu(), ob)
        assert torch.equal(wt[i].cpu(), ow)
        assert torch.allclose(ha[i].cpu(), oha, rtol=1e-6, atol=1e-6)
        # class of the aggregated height may flip only where the mean sits on an integer boundary (sum-order noise)
        assert float((wa[i].cpu() != owa).float().mean()) < 1e-3
    img = torch.rand(2, 8, 64, 64, generator=gen) * 3000 - 200
    mins = np.linspace(-100, 50, 8)
    maxs = mins + np.linspace(1500, 2600, 8)
    out = LM.normalize_tiles(img.to("cuda:0"), mins, maxs, (0, 1))
    want = torch.stack([LO.normalize(img[i], mins, maxs, (0, 1)) for i in range(2)])
    assert torch.allclose(out.cpu(), want, rtol=1e-6, atol=1e-7)
    assert float(out.min()) == 0.0 and float(out.max()) == 1.0
    out2 = LM.normalize_tiles(img.to("cuda:0"), mins, maxs, None)     # grid loader: no clip (BH_loader.py:984-986)
    assert float(out2.min()) < 0.0


def _g13(golden_dir):
    return np.load(os.path.join(golden_dir, "g13_loader.npz"))


def test_loader_oracle_matches_reference_dataset_outputs(golden_dir):
    """g13_loader.npz = outputs of the reference's own myImageFloder_S12_globe.__getitem__ (tools/make_golden.py ran it
    with fake tifffile / cv2 IO): the oracle's restatement must reproduce them (labels bit-exact)."""
    from oracle import loader_oracle as LO
    g = _g13(golden_dir)
    for i in range(g["s2"].shape[0]):
        raw = torch.from_numpy(np.concatenate([g["s2"][i], g["s1"][i]], axis=-1)).permute(2, 0, 1)
        img = LO.normalize(raw, g["mins"], g["maxs"], (0, 1))
        assert torch.allclose(img, torch.from_numpy(g[f"img{i}"]), rtol=1e-6, atol=1e-7)
        hf, ha, b, w, wa = LO.label_prep(g["height_u8"][i], HIR, g["heightweight"])
        assert np.array_equal(ha.numpy(), g[f"height_aggre{i}"]) and np.array_equal(b.numpy(), g[f"build{i}"])
        assert np.array_equal(w.numpy(), g[f"weight{i}"]) and np.array_equal(wa.numpy(), g[f"weight_aggre{i}"])


@pytest.mark.gpu
def test_loader_kernels_match_reference_dataset_outputs(golden_dir):
    from srbh_amd import loader_math as LM
    g = _g13(golden_dir)
    n = g["s2"].shape[0]
    raw = torch.from_numpy(np.concatenate([g["s2"], g["s1"]], axis=-1)).permute(0, 3, 1, 2).contiguous()
    img = LM.normalize_tiles(raw.to("cuda:0"), g["mins"], g["maxs"], (0, 1)).cpu()
    prep = LM.LabelPrep(HIR, g["heightweight"], "cuda:0")
    hf, ha, build, wt, wa = prep(torch.from_numpy(g["height_u8"]).to("cuda:0"))
    for i in range(n):
        assert torch.allclose(img[i], torch.from_numpy(g[f"img{i}"]), rtol=1e-6, atol=1e-7)
        assert np.array_equal(build[i].cpu().numpy(), g[f"build{i}"]) and np.array_equal(wt[i].cpu().numpy(), g[f"weight{i}"])
        assert np.allclose(ha[i].cpu().numpy(), g[f"height_aggre{i}"], rtol=1e-6, atol=1e-6)
        assert float((wa[i].cpu().numpy() != g[f"weight_aggre{i}"]).mean()) < 1e-2
