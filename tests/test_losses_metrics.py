"""Losses and metrics: the oracle against the reference fixtures on CPU, the HIP reductions on GPU."""
import os

import numpy as np
import pytest
import torch

LVS = (0.0, 0.37)


def _g10(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_losses.npz"))
    t = {k: torch.from_numpy(g[k]) for k in g.files}
    t["cls"] = t["cls"].long()
    return t


def test_oracle_losses_match_reference_fixtures(golden_dir):
    from oracle import loss_oracle as LO
    t = _g10(golden_dir)
    for lv in LVS:
        hp = t["height_pred"].clone().requires_grad_(True)
        lvt = torch.tensor(lv, requires_grad=True)
        loss = LO.mse_adapt_weight(hp, t["height"], t["weight"], lvt)
        gi, gl = torch.autograd.grad(loss, [hp, lvt])
        assert torch.allclose(loss, t[f"mse_lv{lv}"], rtol=1e-6)
        assert torch.allclose(gi, t[f"mse_grad_lv{lv}"], rtol=1e-6, atol=1e-9)
        assert torch.allclose(gl, t[f"mse_dlv_lv{lv}"], rtol=1e-6)
        z = t["logits"].clone().requires_grad_(True)
        lvt = torch.tensor(lv, requires_grad=True)
        loss = LO.ce_dice_adapt_weight(z, t["cls"], t["weight"], lvt)
        gi, gl = torch.autograd.grad(loss, [z, lvt])
        assert torch.allclose(loss, t[f"cedice_lv{lv}"], rtol=1e-6)
        assert torch.allclose(gi, t[f"cedice_grad_lv{lv}"], rtol=1e-5, atol=1e-9)
        assert torch.allclose(gl, t[f"cedice_dlv_lv{lv}"], rtol=1e-6)
    assert torch.allclose(LO.mse_adapt_weight(t["height_pred"], t["height"], None, torch.tensor(0.1)), t["mse_unw"], rtol=1e-6)
    assert torch.allclose(LO.ce_dice_adapt_weight(t["logits"], t["cls"], None, torch.tensor(0.1)), t["cedice_unw"], rtol=1e-6)
    assert torch.allclose(LO.dice(t["dice_pred"], t["cls"] > 0), t["dice"], rtol=1e-6)


def test_oracle_metrics_match_reference_fixtures(golden_dir):
    from oracle import loss_oracle as LO
    g = np.load(os.path.join(golden_dir, "g11_metrics.npz"))
    cm = LO.confusion_matrix(torch.from_numpy(g["toy_pred"]), torch.from_numpy(g["toy_ref"]), 3)
    assert np.array_equal(cm.numpy(), g["toy_cm"].astype(np.int64))          # metrics.py:466-469 toy vectors
    cm7 = LO.confusion_matrix(torch.from_numpy(g["seg_pred"]).long(), torch.from_numpy(g["seg_label"]).long(), 7)
    assert np.array_equal(cm7.numpy(), g["seg_cm"].astype(np.int64))
    st = torch.zeros(7, 3, dtype=torch.float64)
    ct = torch.zeros(7, 1, dtype=torch.float64)
    lab = torch.from_numpy(g["seg_label"]).long()
    for b in range(2):
        s_, c_ = LO.height_metric_batch(torch.from_numpy(g["hm_pred"][b]), torch.from_numpy(g["hm_ref"][b]), lab[b], 7)
        st += s_
        ct += c_
    assert np.allclose(st.numpy(), g["hm_stats"], rtol=1e-6) and np.array_equal(ct.numpy(), g["hm_count"])


def test_losses_refuse_cpu_tensors():
    from srbh_amd.losses import _WMSESum
    with pytest.raises(RuntimeError):
        _WMSESum.apply(torch.zeros(4), torch.zeros(4), None)


@pytest.mark.gpu
@pytest.mark.parametrize("channels_last", [False, True])
def test_hip_losses_match_reference_fixtures(golden_dir, channels_last):
    from srbh_amd import losses as SL
    t = {k: v.cuda() for k, v in _g10(golden_dir).items()}
    for lv in LVS:
        hp = t["height_pred"].clone().requires_grad_(True)
        m = SL.MSE_adapt_weight(lv)
        loss = m(hp, t["height"], t["weight"])
        gi, gl = torch.autograd.grad(loss, [hp, m.log_var])
        assert loss.dtype == torch.float32 and m.log_var.is_cuda
        assert torch.allclose(loss, t[f"mse_lv{lv}"], rtol=2e-6)
        assert torch.allclose(gi, t[f"mse_grad_lv{lv}"], rtol=1e-5, atol=1e-9)
        assert torch.allclose(gl, t[f"mse_dlv_lv{lv}"], rtol=2e-6)
        z = t["logits"].clone()
        if channels_last:
            z = z.contiguous(memory_format=torch.channels_last)
        z.requires_grad_(True)
        c = SL.CE_DICE_adapt_weight(lv)
        loss = c(z, t["cls"], t["weight"])
        gi, gl = torch.autograd.grad(loss, [z, c.log_var])
        assert torch.allclose(loss, t[f"cedice_lv{lv}"], rtol=2e-6)
        assert torch.allclose(gi, t[f"cedice_grad_lv{lv}"], rtol=2e-5, atol=2e-9)
        assert torch.allclose(gl, t[f"cedice_dlv_lv{lv}"], rtol=2e-6)
    assert torch.allclose(SL.MSE_adapt(0.1)(t["height_pred"], t["height"]), t["mse_unw"], rtol=2e-6)
    assert torch.allclose(SL.CE_DICE_adapt(0.1)(t["logits"], t["cls"]), t["cedice_unw"], rtol=2e-6)
    assert torch.allclose(SL.Dice()(t["dice_pred"], (t["cls"] > 0)), t["dice"], rtol=2e-6)


@pytest.mark.gpu
def test_hip_losses_full_size_against_oracle():
    """(B,7,256,256) logits / (B,256,256) heights: the HIP sums against the oracle on the host (linearity in the
    weights is the size-independent property: loss(w1) + loss(w2) - 2*log_var terms == loss(w1 + w2))."""
    from oracle import loss_oracle as LO
    from srbh_amd import losses as SL
    g = torch.Generator()
    g.manual_seed(77)
    B = 4
    z = torch.randn(B, 7, 256, 256, generator=g) * 2
    y = torch.randint(0, 7, (B, 256, 256), generator=g)
    w = torch.rand(B, 256, 256, generator=g)
    hp, ht = torch.rand(B, 256, 256, generator=g) * 60, torch.rand(B, 256, 256, generator=g) * 60
    lv = torch.tensor(0.2)
    c = SL.CE_DICE_adapt_weight(0.2)
    zc = z.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    got = c(zc, y.cuda(), w.cuda())
    (gz,) = torch.autograd.grad(got, [zc])
    zr = z.clone().requires_grad_(True)
    want = LO.ce_dice_adapt_weight(zr, y, w, lv)
    (gr,) = torch.autograd.grad(want, [zr])
    assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want))
    assert float((gz.cpu() - gr).norm() / gr.norm()) <= 1e-5
    # extreme logits: the target class 200 below the maximum.  exp(z_y - max) underflows, the log-softmax form stays
    # finite exactly like nn.CrossEntropyLoss (reference losses_pytorch/selfloss.py:145-168)
    ze = torch.zeros(1, 7, 8, 8)
    ze[:, 3] = 200.0
    ye = torch.zeros(1, 8, 8, dtype=torch.long)
    zec = ze.cuda().requires_grad_(True)
    got_e = SL.CE_DICE_adapt_weight(0.0)(zec, ye.cuda(), torch.ones(1, 8, 8).cuda())
    want_e = LO.ce_dice_adapt_weight(ze, ye, torch.ones(1, 8, 8), torch.tensor(0.0))
    assert bool(torch.isfinite(got_e)) and abs(float(got_e) - float(want_e)) <= 1e-5 * abs(float(want_e))
    (ge,) = torch.autograd.grad(got_e, [zec])
    assert bool(torch.isfinite(ge).all())
    yb = ye.clone()
    yb[0, 0, 0] = 7                                    # out-of-range label: the reference raises, the kernel poisons
    assert bool(torch.isnan(SL.CE_DICE_adapt_weight(0.0)(ze.cuda(), yb.cuda(), torch.ones(1, 8, 8).cuda())))
    m = SL.MSE_adapt_weight(0.0)
    a = m(hp.cuda(), ht.cuda(), w.cuda())
    b = m(hp.cuda(), ht.cuda(), (1 - w).cuda())
    u = SL.MSE_adapt(0.0)(hp.cuda(), ht.cuda())
    assert abs(float(a + b) - float(u)) <= 2e-6 * float(u)
    assert abs(float(a) - float(LO.mse_adapt_weight(hp, ht, w, torch.tensor(0.0)))) <= 2e-6 * float(a)


@pytest.mark.gpu
def test_hip_metrics_match_reference_fixtures(golden_dir):
    from srbh_amd import metrics as SM
    g = np.load(os.path.join(golden_dir, "g11_metrics.npz"))
    m = SM.SegmentationMetric(3, "cuda")
    m.addBatch(torch.from_numpy(g["toy_pred"]).cuda(), torch.from_numpy(g["toy_ref"]).cuda())
    assert np.array_equal(m.confusionMatrix.cpu().numpy(), g["toy_cm"])      # integer-exact
    for name, fn in (("toy_fwiou", m.Frequency_Weighted_Intersection_over_Union), ("toy_oa", m.OverallAccuracy),
                     ("toy_precision", m.Precision), ("toy_recall", m.Recall), ("toy_f1", m.F1score),
                     ("toy_iou", m.IntersectionOverUnion), ("toy_miou", m.meanIntersectionOverUnion), ("toy_mfwiou", m.mFWIoU)):
        assert np.allclose(fn().cpu().numpy(), g[name], rtol=1e-12, equal_nan=True), name
    m7 = SM.SegmentationMetric(7, "cuda")
    lab = torch.from_numpy(g["seg_label"]).long().cuda()
    prd = torch.from_numpy(g["seg_pred"]).long().cuda()
    m7.addBatch(prd, lab)
    assert np.array_equal(m7.confusionMatrix.cpu().numpy(), g["seg_cm"])
    _, bad = m7.genConfusionMatrix(prd + 7, lab)                              # out-of-range predictions are flagged
    assert int(bad) == 1
    m7.OverallAccuracy()                                                      # (genConfusionMatrix alone is not sticky)
    m7b = SM.SegmentationMetric(7, "cuda")
    m7b.addBatch(prd + 7, lab)                                                # ... addBatch is: reading a score raises
    with pytest.raises(ValueError):
        m7b.OverallAccuracy()
    with pytest.raises(ValueError):
        m7b.getConfusionMatrix()
    hm = SM.HeightMetric(7, "cuda")
    tref = torch.tensor([0, 0, 3, 6, 5, 1]).float().cuda()
    tpred = torch.tensor([0, 1, 0, 1, 0, 2]).float().cuda()
    hm.addBatch(tpred, tref, tref)
    assert np.allclose(hm.stats.cpu().numpy(), g["hm_toy_stats"], rtol=1e-6)
    assert np.array_equal(hm.count.cpu().numpy(), g["hm_toy_count"])
    assert np.allclose(hm.getAvgAll().cpu().numpy(), g["hm_toy_all"], rtol=1e-6)
    hm2 = SM.HeightMetric(7, "cuda")
    for b in range(2):
        hm2.addBatch(torch.from_numpy(g["hm_pred"][b]).cuda(), torch.from_numpy(g["hm_ref"][b]).cuda(), lab[b])
    assert np.allclose(hm2.stats.cpu().numpy(), g["hm_stats"], rtol=1e-6)
    assert np.array_equal(hm2.count.cpu().numpy(), g["hm_count"])
    assert np.allclose(hm2.getAvgEach().cpu().numpy(), g["hm_each"], rtol=1e-6)
    assert np.allclose(hm2.getAvgBalance().cpu().numpy(), g["hm_balance"], rtol=1e-6)


@pytest.mark.gpu
def test_confusion_full_size_sums_to_pixel_count():
    from srbh_amd import metrics as SM
    g = torch.Generator()
    g.manual_seed(3)
    lab = torch.randint(0, 7, (16, 256, 256), generator=g).cuda()
    prd = torch.randint(0, 7, (16, 256, 256), generator=g).cuda()
    m = SM.SegmentationMetric(7, "cuda")
    m.addBatch(prd, lab)
    want = torch.bincount((7 * lab.flatten() + prd.flatten()).cpu(), minlength=49).reshape(7, 7).double()
    assert torch.equal(m.confusionMatrix.cpu(), want)
    assert float(m.confusionMatrix.sum()) == 16 * 256 * 256
