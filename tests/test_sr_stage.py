"""SURVEY.md 8f-4, first slice (SR-stage fine-tuning): the generator's forward + backward on libsrbh against the reference's own
autograd (fixture g14, produced by tools/make_golden.py from /root/reference), the stock-op companions (discriminator,
filter2D, GANLoss) against the same fixture on the CPU, and one RealESRGAN(is_train=True) step on the device."""
import os

import numpy as np
import pytest
import torch

from oracle import srbh_oracle as O
from oracle import synth


def _g14(golden_dir):
    return np.load(os.path.join(golden_dir, "g14_sr_stage.npz"))


def rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def test_sr_companions_match_reference_fixture(golden_dir):
    from srbh_amd.srgan import GANLoss, UNetDiscriminatorSN, filter2D
    g = _g14(golden_dir)
    d = UNetDiscriminatorSN(3, num_feat=8, skip_connection=True).eval()
    sd = {k[len("disc_sd_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("disc_sd_")}
    d.load_state_dict(sd, strict=True)                        # same keys as the reference module (spectral-norm buffers included)
    with torch.no_grad():
        y = d(rand((2, 3, 32, 32), 143, 0.0, 1.0))
    assert O.rel_l2(y, torch.from_numpy(g["disc_out"])) <= 1e-6
    img = rand((2, 3, 20, 24), 144, 0.0, 1.0)
    k1, kb = rand((1, 5, 5), 145, 0.0, 1.0), rand((2, 7, 7), 146, 0.0, 1.0)
    assert O.rel_l2(filter2D(img, k1 / k1.sum()), torch.from_numpy(g["f2d_shared"])) <= 1e-6
    assert O.rel_l2(filter2D(img, kb / kb.sum(dim=(1, 2), keepdim=True)), torch.from_numpy(g["f2d_batch"])) <= 1e-6
    z = rand((2, 1, 8, 8), 147, -2.0, 2.0)
    for t in ("vanilla", "lsgan", "wgan", "wgan_softplus", "hinge"):
        gl = GANLoss(t, loss_weight=0.1)
        got = torch.stack([gl(z, True, is_disc=False), gl(z, True, is_disc=True), gl(z, False, is_disc=True)])
        assert torch.allclose(got, torch.from_numpy(g[f"gan_{t}"]), rtol=1e-6, atol=1e-7), t
    with pytest.raises(NotImplementedError):
        GANLoss("nope")


def test_usm_sharp_gaussian_kernel_known_answers():
    """cv2.getGaussianKernel(51, 0) restated (cv2 is absent offline): sigma = 0.3*((51-1)*0.5 - 1) + 0.8 = 8.0; the kernel is the
    normalised outer product, symmetric, and a constant image is a fixed point of the sharpener."""
    from srbh_amd.srgan import USMSharp, _gaussian_kernel_1d
    g = _gaussian_kernel_1d(51, 0)
    assert abs(float(g.sum()) - 1.0) < 1e-6 and torch.allclose(g, g.flip(0))
    assert abs(float(g[25] / g[24]) - float(np.exp(1.0 / (2 * 8.0 ** 2)))) < 1e-6      # exp(-(0)^2/2s^2) / exp(-(1)^2/2s^2), s = 8
    u = USMSharp()
    assert u.radius == 51 and tuple(u.kernel.shape) == (1, 51, 51) and abs(float(u.kernel.sum()) - 1) < 1e-5
    flat = torch.full((1, 3, 64, 64), 0.37)
    assert torch.allclose(u(flat), flat, atol=1e-6)
    x = rand((1, 3, 64, 64), 5, 0.0, 1.0)
    y = u(x)
    assert y.shape == x.shape and float(y.min()) >= -1e-6 and float(y.max()) <= 1 + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["fw", "ft"])
def test_rrdbnet_backward_matches_reference_autograd(golden_dir, tag):
    """forward() / forward_feature() of a 2-block RRDBNet with a recorded graph: outputs (<= 1e-5), input gradient and every
    parameter gradient against the reference's autograd.  Gradient tolerance 2e-3: LeakyReLU's derivative jumps at 0, and
    among the 10^6 activations of this net ONE pre-activation of conv_hr is 8e-7 -- its sign (hence a factor 5 on that
    element's gradient) depends on the fp32 summation order; that single element moves every upstream gradient by ~4e-4
    (measured against float64 autograd: the reference's own fp32 run lands on the float64 side, this kernel on the other).
    The kernels themselves are exact: test_training_convs_exact_against_float64 below."""
    from srbh_amd.rrdbnet import RRDBNet
    g = _g14(golden_dir)
    sd = synth.rrdbnet_state_dict(num_block=2, seed=31, mode="stress")
    net = RRDBNet(3, 3, num_block=2)
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").train().enable_training_path(True)
    x = rand((2, 3, 16, 16), 140, 0.0, 1.0).to("cuda:0").requires_grad_(True)
    y = net(x) if tag == "fw" else net.forward_feature(x)
    w = rand(tuple(y.shape), 141 if tag == "fw" else 142).to("cuda:0")
    (y * w).sum().backward()
    want_out = torch.from_numpy(g[f"{tag}_out"])
    got_out = y.detach().cpu() if tag == "fw" else y.detach().cpu()[:, ::8, ::4, ::4]
    assert O.rel_l2(got_out, want_out) <= 1e-5
    GTOL = 2e-3
    assert O.rel_l2(x.grad.cpu(), torch.from_numpy(g[f"{tag}_gx"])) <= GTOL
    names = [str(n) for n in g["param_names"]]
    params = dict(net.named_parameters())
    assert names == list(params.keys())
    gn = g[f"{tag}_gnorm"]
    for i, k in enumerate(names):
        p = params[k]
        if gn[i] == 0.0:
            assert p.grad is None or float(p.grad.norm()) == 0.0, k       # conv_last is outside forward_feature's graph
        else:
            assert abs(float(p.grad.double().norm()) - gn[i]) <= GTOL * gn[i], k
    for key in g.files:
        if key.startswith(f"{tag}_g_"):
            k = key[len(f"{tag}_g_"):]
            got = params[k].grad.cpu()
            got = got if got.numel() <= 4096 else got[::4, ::4]
            assert O.rel_l2(got, torch.from_numpy(g[key])) <= GTOL, k


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,hw", [(64, 3, 64), (64, 64, 32), (192, 64, 16), (96, 32, 16), (160, 32, 24), (3, 64, 16)])
def test_training_convs_exact_against_float64(cin, cout, hw):
    """the three kernels of the training path -- forward conv, data gradient (transposed + flipped packs, 64 input channels per
    launch, accumulated in place), weight gradient from a STRIDED view of a 192-channel dense buffer -- against float64 torch
    ops on the device: <= 2e-6 (fp32 summation order only)."""
    import torch.nn.functional as F
    from srbh_amd import rrdbnet_autograd as A
    torch.manual_seed(cin * 1000 + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).cuda()
    D = torch.randn(2, hw, hw, 192, device="cuda")                 # the dense buffer; the conv sees its first `cin` channels
    x = D[..., :cin] if cin <= 192 else None
    g = torch.randn(2, hw, hw, cout, device="cuda")
    xd = x.permute(0, 3, 1, 2).double().contiguous().requires_grad_(True)
    wd = conv.weight.double().detach().requires_grad_(True)
    y = F.conv2d(xd, wd, conv.bias.double(), 1, 1)
    y.backward(g.permute(0, 3, 1, 2).double())
    p = A._packs(conv)
    out = torch.empty(2, hw, hw, cout, device="cuda")
    A._conv(D, cin, 192, p.fwd, p.bias, cout, out)
    assert O.rel_l2(out.permute(0, 3, 1, 2).cpu(), y.detach().cpu()) <= 2e-6
    acc = cin % 4 == 0                                             # (conv_first's 3-channel input gradient is never accumulated)
    dD = torch.ones(2, hw, hw, 192, device="cuda")                 # accumulate on top of ones: checks the in-place residual epilogue
    A._dgrad_into(g, cout, p, dD, 192, acc)
    assert O.rel_l2((dD[..., :cin] - (1 if acc else 0)).permute(0, 3, 1, 2).cpu(), xd.grad.cpu()) <= 2e-6
    assert bool((dD[..., cin:] == 1).all())                        # nothing outside the conv's input channels was touched
    assert O.rel_l2(A._wgrad(D, cin, 192, g, cout).cpu(), wd.grad.cpu()) <= 2e-6


@pytest.mark.gpu
def test_discriminator_on_the_device_matches_the_reference_fixture(golden_dir):
    """UNetDiscriminatorSN (SR/rrdbnet_arch.py:244-303) as the SR-stage trainer runs it -- on the ROCm device -- against the output the
    imported reference produced on the CPU (g14 `disc_out`): forward <= 1e-5, and its loss-dict protocol keys in the reference's order."""
    from srbh_amd.srgan import UNetDiscriminatorSN
    g = _g14(golden_dir)
    d = UNetDiscriminatorSN(3, num_feat=8, skip_connection=True).eval()
    d.load_state_dict({k[len("disc_sd_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("disc_sd_")}, strict=True)
    d = d.to("cuda:0")
    with torch.no_grad():
        y = d(rand((2, 3, 32, 32), 143, 0.0, 1.0).to("cuda:0"))
    assert O.rel_l2(y.cpu(), torch.from_numpy(g["disc_out"])) <= 1e-5


def _vgg19_seeded_state_dict():
    """the weights tools/make_golden.py gave the reference's feature stack (torch.manual_seed(1905), default init x 3), torchvision's keys"""
    from srbh_amd import srgan
    torch.manual_seed(1905)
    f = srgan.vgg19_features()
    with torch.no_grad():
        for p_ in f.parameters():
            p_.mul_(3.0)
    return {"features." + k: v.clone() for k, v in f.state_dict().items()}


def test_perceptual_loss_matches_the_reference_fixture(golden_dir):
    """srgan.VGGFeatureExtractor / PerceptualLoss (SR/srloss.py:61-143) against fixture g15: the reference's own PerceptualLoss executed on a VGG19
    feature stack built from its layer table with seeded weights (torchvision is absent offline) -- the list form with the default cuts
    [2, 7, 16, 25, 34] / weights / ImageNet input norm / L1, and the single-cut MSE form with the range map: loss values, input gradients,
    feature shapes and norms; parameter keys as upstream; frozen parameters."""
    from srbh_amd.srgan import PerceptualLoss
    g = np.load(os.path.join(golden_dir, "g15_perceptual.npz"))
    sd = _vgg19_seeded_state_dict()
    x = rand((2, 3, 48, 40), 150, 0.0, 1.0).requires_grad_(True)
    gt = rand((2, 3, 48, 40), 151, 0.0, 1.0)
    pl = PerceptualLoss(loss_weight=1.0, use_input_norm=True, use_range_norm=False, state_dict=sd)
    assert not any(p_.requires_grad for p_ in pl.parameters())
    assert [k for k, _ in pl.vgg.features.named_parameters()][:3] == ["child0.0.weight", "child0.0.bias", "child0.2.weight"]
    l = pl(x, gt)
    l.backward()
    assert abs(float(l) - float(g["l1_list_loss"])) <= 1e-5 * abs(float(g["l1_list_loss"]))
    assert O.rel_l2(x.grad, torch.from_numpy(g["l1_list_gx"])) <= 1e-5
    feats = pl.vgg(x.detach())
    assert [list(f.shape) for f in feats] == g["feat_shapes"].tolist()
    assert np.allclose([float(f.double().norm()) for f in feats], g["feat_norms"], rtol=1e-5)
    x.grad = None
    pm = PerceptualLoss(feature_layer=7, lossfn_type="l2", use_input_norm=False, use_range_norm=True, loss_weight=0.5, state_dict=sd)
    assert pm.loss_weight == 1.0          # (pinned upstream, srloss.py:123)
    l2 = pm(x * 2 - 1, gt * 2 - 1)
    l2.backward()
    assert abs(float(l2) - float(g["mse_single_loss"])) <= 1e-5 * abs(float(g["mse_single_loss"]))
    assert O.rel_l2(x.grad, torch.from_numpy(g["mse_single_gx"])) <= 1e-5


def test_one_band_generator_takes_a_three_band_checkpoint_averaged(tmp_path):
    """SR/rrdbnet_arch.py:451-455,470-474: in_ch == 1 -> conv_first.weight averaged over its input bands, conv_last.weight / .bias over its output
    bands, before load_state_dict"""
    from srbh_amd.rrdbnet import RRDBNet, RealESRGAN
    torch.manual_seed(4)
    src = RRDBNet(3, 3, num_block=1)
    path = os.path.join(tmp_path, "net_g.tar")
    torch.save({"params_ema": src.state_dict()}, path)
    m = RealESRGAN(in_ch=1, out_ch=1, num_block=1, device="cpu", pretrain_g_path=path, is_train=False)
    sd, got = src.state_dict(), m.net_g.state_dict()
    assert tuple(got["conv_first.weight"].shape) == (64, 1, 3, 3) and tuple(got["conv_last.weight"].shape) == (1, 64, 3, 3)
    assert torch.equal(got["conv_first.weight"], sd["conv_first.weight"].mean(dim=1, keepdim=True))
    assert torch.equal(got["conv_last.weight"], sd["conv_last.weight"].mean(dim=0, keepdim=True))
    assert torch.equal(got["conv_last.bias"], sd["conv_last.bias"].mean(dim=0, keepdim=True))
    assert torch.equal(got["body.0.rdb1.conv1.weight"], sd["body.0.rdb1.conv1.weight"])


@pytest.mark.gpu
def test_realesrgan_training_step_with_the_perceptual_term():
    """RealESRGAN(is_train=True, vgg19_weights=<state_dict>): the VGG19 perceptual term joins the generator's loss (SR/rrdbnet_arch.py:559-562),
    loss_dict carries the reference's keys in its order, gradients reach the generator through the frozen VGG stack"""
    from srbh_amd.rrdbnet import RealESRGAN
    torch.manual_seed(3)
    m = RealESRGAN(3, 3, num_block=1, device="cuda:0", is_train=True, vgg19_weights=_vgg19_seeded_state_dict())
    gt = torch.nn.functional.interpolate(rand((2, 3, 16, 16), 9, 0.0, 1.0), scale_factor=8, mode="bilinear")
    lq = torch.nn.functional.avg_pool2d(gt, 4)
    vals = []
    for it in range(4):
        m.feed_data({"lq": lq, "gt": gt})
        ld = m.optimize_parameters()
        assert list(ld) == ["l_g_pix", "l_g_percep", "l_g_gan", "l_d_real", "out_d_real", "l_d_fake", "out_d_fake"]
        vals.append(ld["l_g_percep"])
    assert all(v == v and v > 0 for v in vals) and vals[-1] < vals[0], vals


@pytest.mark.gpu
def test_realesrgan_training_step_runs_and_learns():
    """RealESRGAN(is_train=True) (reference SR/rrdbnet_arch.py:437-592): feed_data -> optimize_parameters for a few iterations on
    one synthetic LR/HR pair: the pixel loss goes down, the EMA copy moves, the discriminator trains, save() writes both nets."""
    import tempfile
    from srbh_amd.rrdbnet import RealESRGAN
    torch.manual_seed(3)
    m = RealESRGAN(3, 3, num_block=1, device="cuda:0", is_train=True)
    assert m.net_g.training and hasattr(m, "net_g_ema") and not any(p.requires_grad for p in m.net_g_ema.parameters())
    gt = torch.nn.functional.interpolate(rand((2, 3, 16, 16), 9, 0.0, 1.0), scale_factor=8, mode="bilinear")   # smooth 128x128 target
    lq = torch.nn.functional.avg_pool2d(gt, 4)
    ema0 = m.net_g_ema.conv_first.weight.clone()
    losses = []
    for it in range(8):
        m.feed_data({"lq": lq, "gt": gt})
        ld = m.optimize_parameters()
        m.update_learning_rate(it)
        losses.append(ld["l_g_pix"])
        assert list(ld) == ["l_g_pix", "l_g_gan", "l_d_real", "out_d_real", "l_d_fake", "out_d_fake"]     # (the reference's keys, its order; no perceptual plug-in)
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    assert not torch.equal(m.net_g_ema.conv_first.weight, ema0)
    with tempfile.TemporaryDirectory() as td:
        m.save(0, 8, td)
        ck = torch.load(os.path.join(td, "net_g.tar"), map_location="cpu")
        assert set(ck) >= {"params", "params_ema", "epoch", "current_iter"} and len(ck["params"]) == len(m.net_g.state_dict())
    with torch.no_grad():
        assert m.predict(lq).shape == (2, 3, 128, 128)


@pytest.mark.gpu
def test_realesrgan_training_steps_in_fast_precision_learn_and_repack():
    """The SR-stage step with the generator's dense blocks on the trunk's kernel family (set_train_precision("fast")): 64 x 64 LR tiles
    (the geometry that path is written for), Adam updates between the steps -- so the per-RDB forward packs AND the stacked bf16
    gradient packs must be rebuilt every step (cache keys: version, optimizer stamp) -- the pixel loss falls, and the first step's
    generator gradients agree with the exact-fp32 graph's (cosine >= 0.99 on the dense-block weights)."""
    from srbh_amd import rrdbnet_autograd as RA
    from srbh_amd.rrdbnet import RealESRGAN
    gt = torch.nn.functional.interpolate(rand((2, 3, 32, 32), 9, 0.0, 1.0), scale_factor=8, mode="bilinear")   # smooth 256x256 target
    lq = torch.nn.functional.avg_pool2d(gt, 4)                                                                    # 64x64
    grads = {}
    try:
        for mode in ("f32", "fast"):
            RA.set_train_precision(mode)
            RA._FAST_WS.clear()
            torch.manual_seed(3)
            m = RealESRGAN(3, 3, num_block=1, device="cuda:0", is_train=True)
            losses = []
            for it in range(6 if mode == "fast" else 1):
                m.feed_data({"lq": lq, "gt": gt})
                if it == 0:
                    m.optimizer_g.zero_grad()
                    m.cri_pix(m.net_g(m.lq), m.gt_usm).backward()
                    grads[mode] = {k: p.grad.detach().clone() for k, p in m.net_g.named_parameters() if p.grad is not None}
                losses.append(m.optimize_parameters()["l_g_pix"])
            if mode == "fast":
                assert RA._FAST_WS, "the fast trunk path did not run"
                assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    finally:
        RA.set_train_precision("f32")
    cos = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()).clamp_min(1e-300))   # noqa: E731
    body = [k for k in grads["f32"] if k.startswith("body.") and k.endswith("weight")]
    assert len(body) == 15 and all(cos(grads["fast"][k], grads["f32"][k]) >= 0.99 for k in body), {k: cos(grads["fast"][k], grads["f32"][k]) for k in body}


@pytest.mark.gpu
@pytest.mark.parametrize("fast_mode", ["mixed", "fast"])
def test_rrdbnet_mixed_precision_training_graph_close_to_exact(fast_mode):
    """("fast", round 3: the dense blocks forward AND backward on the trunk's own 32x32x16-MFMA kernel family -- ACT16 planes, fp16
    forward, bf16 gradient convs with the LeakyReLU mask folded in, weight gradients straight from the saved planes -- same bounds;
    64 x 64 tiles, the geometry that kernel family is written for.)
    rrdbnet_autograd.set_train_precision("mixed"): forward convs with fp16 operands, data / weight gradients with bf16 operands
    (fp32 accumulation, fp32 residual and LeakyReLU epilogues).  Against the exact-fp32 graph of the same 2-block net: output within
    1e-3 (measured 3.6e-4 for both modes, the inference trunk's own 3.3e-4 level: tools/sr_precision_probe.py), every parameter gradient with cosine >= 0.995 and norm within 5 %,
    and the mixed graph is not slower beyond noise (its speed is bench.py --workload sr_train's number)."""
    import time
    from srbh_amd import rrdbnet_autograd as RA
    from srbh_amd.rrdbnet import RRDBNet
    sd = synth.rrdbnet_state_dict(num_block=2, seed=31, mode="stress")
    res, times = {}, {}
    try:
        hw = 64 if fast_mode == "fast" else 32
        for mode in ("f32", fast_mode):
            RA.set_train_precision(mode)
            net = RRDBNet(3, 3, num_block=2)
            net.load_state_dict(sd, strict=True)
            net = net.to("cuda:0").train().enable_training_path(True)
            x = rand((2, 3, hw, hw), 140, 0.0, 1.0).to("cuda:0").requires_grad_(True)
            w = None
            for rep in range(3):
                for p in net.parameters():
                    p.grad = None
                x.grad = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                y = net(x)
                if w is None:
                    w = rand(tuple(y.shape), 141).to("cuda:0")
                (y * w).sum().backward()
                torch.cuda.synchronize()
                times[mode] = min(times.get(mode, 1e9), time.perf_counter() - t0)
            res[mode] = (y.detach().cpu(), x.grad.cpu(), {k: p.grad.cpu() for k, p in net.named_parameters()})
    finally:
        RA.set_train_precision("f32")
    (y0, gx0, g0), (y1, gx1, g1) = res["f32"], res[fast_mode]
    if fast_mode == "fast":
        assert RA._FAST_WS, "the fast trunk path did not run"
    assert 1e-6 < O.rel_l2(y1, y0) <= 1e-3
    cos = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()).clamp_min(1e-300))
    assert cos(gx1, gx0) >= 0.995
    for k in g0:
        assert cos(g1[k], g0[k]) >= 0.995, k
        assert abs(float(g1[k].norm()) / float(g0[k].norm()) - 1.0) <= 0.05, k
    # (wall clock of a 2-block net on 2 tiles is launch-bound and noisy: this only guards against a pathologically slow path; the speed
    #  itself is measured by `bench.py --workload sr_train`, which reports both modes)
    assert times[fast_mode] < 1.5 * times["f32"], times


@pytest.mark.gpu
def test_fast_mode_two_forwards_before_backward_keep_their_own_saved_planes():
    """Round-3 ADVICE (medium): the fast mode's saved activation planes lived in ONE module-global buffer per geometry, so a second
    forward at the same geometry before the first backward (two generator passes, a second net) overwrote them and the first backward
    returned wrong gradients silently.  Now every forward leases its own buffer set until its backward: the gradients of forward A,
    taken AFTER an interleaved forward B on other data, equal the gradients of A run alone."""
    from srbh_amd import rrdbnet_autograd as RA
    from srbh_amd.rrdbnet import RRDBNet
    sd = synth.rrdbnet_state_dict(num_block=1, seed=33, mode="stress")
    try:
        RA.set_train_precision("fast")
        net = RRDBNet(3, 3, num_block=1)
        net.load_state_dict(sd, strict=True)
        net = net.to("cuda:0").train().enable_training_path(True)
        xa = rand((2, 3, 64, 64), 150, 0.0, 1.0).to("cuda:0")
        xb = rand((2, 3, 64, 64), 151, 0.0, 1.0).to("cuda:0")

        def grads_of(interleave):
            for p in net.parameters():
                p.grad = None
            ya = net(xa)
            if interleave:
                yb = net(xb)                       # same geometry, other data, before A's backward
            ya.square().sum().backward()
            g = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
            if interleave:
                for p in net.parameters():
                    p.grad = None
                yb.square().sum().backward()       # B's own planes are intact too
                gb = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
                return g, gb
            return g

        alone_a = grads_of(False)
        inter_a, inter_b = grads_of(True)
        xa, xb = xb, xa
        alone_b = grads_of(False)
        pool = next(iter(RA._FAST_WS.values()))
        assert len(pool) == 2 and not any(ws["busy"] for ws in pool)          # the overlap cost one extra set; both are free again
        for k in alone_a:
            assert torch.allclose(inter_a[k], alone_a[k], rtol=1e-4, atol=1e-6 * float(alone_a[k].abs().max())), k
            assert torch.allclose(inter_b[k], alone_b[k], rtol=1e-4, atol=1e-6 * float(alone_b[k].abs().max())), k
    finally:
        RA.set_train_precision("f32")


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,B", [(1, 2), (2, 3)])
def test_persistent_training_forward_equals_the_per_layer_sequence(blocks, B):
    """SR-stage "fast" training forward of the trunk (SR/rrdbnet_arch.py:136-167 per RDB) as ONE launch of the inference trunk's persistent kernel
    over a row of dense buffers (srbh_rrdbnet_trunk_train_forward_persistent) against the per-layer sequence: the fp32 output and EVERY saved
    plane of every RDB (what the backward reads: LeakyReLU masks, weight-gradient operands) bit-identical."""
    from srbh_amd import rrdbnet_autograd as RA
    from srbh_amd import synth
    from srbh_amd.rrdbnet import RRDBNet
    net = RRDBNet(3, 3, num_block=blocks)
    net.load_state_dict(synth.rrdbnet_state_dict(num_block=blocks, seed=5, mode="stress"))
    net = net.to("cuda:0")
    feat = rand((B, 64, 64, 64), 31).to("cuda:0").contiguous()          # (B, H, W, 64) fp32 NHWC
    outs = []
    for persistent in (True, False):
        RA._FAST_WS.clear()
        n0 = dict(RA.TRUNK_FWD_PATHS)
        ws = RA._fast_buffers(B, 64, 64, blocks * 3, feat.device)
        if not persistent:
            ws["aux"] = None
        xr, lease = RA._trunk_fast_forward(net, feat)
        torch.cuda.synchronize()
        assert RA.TRUNK_FWD_PATHS["persistent" if persistent else "per_layer"] == n0["persistent" if persistent else "per_layer"] + 1
        outs.append((xr.clone(), lease.ws["D"].clone()))
        lease.release()
    RA._FAST_WS.clear()
    assert bool(torch.isfinite(outs[0][0]).all())
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,B", [(1, 2), (2, 3), (1, 33)])          # (33 images: two launches of the persistent kernel, one workgroup per CU)
def test_persistent_training_backward_follows_the_per_layer_sequence(blocks, B):
    """SR-stage "fast" training backward of the trunk (the gradient of SR/rrdbnet_arch.py:136-167 per RDB, SR/rrdbnet_arch.py:538-592's
    l_g_total.backward()): the data-gradient convs of all RDBs as ONE launch of the persistent kernel's bf16 form
    (srbh_rrdbnet_trunk_train_backward_persistent) against the per-layer sequence on the SAME saved planes and the same output gradient.  Same bf16
    operands and summation order per conv; the fp32 gradient streams differ in their last bits (0.2 applied to conv5's sum instead of the stream),
    which moves a bf16 rounding here and there: input gradient and every weight / bias gradient within 2e-3 rel-L2 (measured ~1e-4)."""
    from srbh_amd import rrdbnet_autograd as RA
    from srbh_amd import synth
    from srbh_amd.rrdbnet import RRDBNet
    net = RRDBNet(3, 3, num_block=blocks)
    net.load_state_dict(synth.rrdbnet_state_dict(num_block=blocks, seed=5, mode="stress"))
    net = net.to("cuda:0")
    feat = rand((B, 64, 64, 64), 31).to("cuda:0").contiguous()
    g = rand((B, 64, 64, 64), 32, -1.0, 1.0).to("cuda:0").contiguous()
    res = []
    old = RA.BWD_PERSISTENT
    try:
        for persistent in (True, False):
            RA.BWD_PERSISTENT = True          # (both runs on the row of G buffers: only the launch form differs)
            RA._FAST_WS.clear()
            xr, lease = RA._trunk_fast_forward(net, feat)
            RA.BWD_PERSISTENT = persistent
            n0 = dict(RA.TRUNK_BWD_PATHS)
            grads = {}
            gin = RA._trunk_fast_backward(net, lease, g.clone(), grads)
            torch.cuda.synchronize()
            key = "persistent" if persistent else "per_layer"
            assert RA.TRUNK_BWD_PATHS[key] == n0[key] + 1
            lease.release()
            named = {n_: grads[id(p)].clone() for n_, p in net.named_parameters() if id(p) in grads}
            res.append((gin.clone(), named))
    finally:
        RA.BWD_PERSISTENT = old
        RA._FAST_WS.clear()
    (g0, w0), (g1, w1) = res
    assert bool(torch.isfinite(g0).all()) and len(w0) == blocks * 3 * 10 and w0.keys() == w1.keys()
    assert O.rel_l2(g0.cpu(), g1.cpu()) <= 2e-3
    worst = max(O.rel_l2(w0[k].cpu(), w1[k].cpu()) for k in w0)
    assert worst <= 2e-3, worst


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,B,Hh", [(1, 2, 64), (2, 5, 64), (1, 1, 128)])
def test_one_launch_trunk_weight_gradients_equal_the_general_kernel(blocks, B, Hh):
    """srbh_trunk_wgrad (weight + bias gradients of every RDB's five convs, SR/rrdbnet_arch.py:136-167, as one launch over (RDB, plane pair, tile
    range) + one ordered reduce) against the general 16 x 16-block kernel called RDB by RDB on the SAME saved planes and gradient planes: same bf16
    operands, fp32 accumulation in a different order -- every gradient within 2e-5 rel-L2 (B = 5: a tile count the four splits do not divide)."""
    from srbh_amd import rrdbnet_autograd as RA
    from srbh_amd import synth
    from srbh_amd.rrdbnet import RRDBNet
    net = RRDBNet(3, 3, num_block=blocks)
    net.load_state_dict(synth.rrdbnet_state_dict(num_block=blocks, seed=5, mode="stress"))
    net = net.to("cuda:0")
    feat = rand((B, Hh, 64, 64), 31).to("cuda:0").contiguous()          # (B, H, W, 64): 64 pixels wide, H a multiple of 8
    g = rand((B, Hh, 64, 64), 32, -1.0, 1.0).to("cuda:0").contiguous()
    res = []
    old = RA.TRUNK_WGRAD
    try:
        for one_launch in (True, False):
            RA._FAST_WS.clear()
            xr, lease = RA._trunk_fast_forward(net, feat)
            RA.TRUNK_WGRAD = one_launch
            grads = {}
            gin = RA._trunk_fast_backward(net, lease, g.clone(), grads)
            torch.cuda.synchronize()
            lease.release()
            res.append((gin.clone(), {n_: grads[id(p)].clone() for n_, p in net.named_parameters() if id(p) in grads}))
    finally:
        RA.TRUNK_WGRAD = old
        RA._FAST_WS.clear()
    assert RA.TRUNK_BWD_PATHS["persistent"] >= 2          # (both runs took the persistent data-gradient launch: only the weight gradients differ)
    (g0, w0), (g1, w1) = res
    assert torch.equal(g0, g1) and len(w0) == blocks * 3 * 10
    for k in w0:
        assert bool(torch.isfinite(w0[k]).all()) and float(w0[k].abs().max()) > 0, k
        assert O.rel_l2(w0[k].cpu(), w1[k].cpu()) <= 2e-5, (k, O.rel_l2(w0[k].cpu(), w1[k].cpu()))


@pytest.mark.gpu
def test_usm_sharp_on_the_device_is_the_two_dimensional_form():
    """USMSharp (SR/rrdbnet_arch.py:412-434) on a device tensor applies the 51 x 51 Gaussian as its two 1-D factors; against the 2-D form on the
    CPU (what the reference fixture pins): the blur itself to fp32 summation order, the sharpened image up to the rare pixel whose residual sits
    on the threshold (its mask flips and the blurred mask spreads ~1e-3 of it)."""
    from srbh_amd.srgan import USMSharp, filter2D
    u = USMSharp()
    img = torch.nn.functional.interpolate(rand((2, 3, 40, 40), 5, 0.0, 1.0), scale_factor=4, mode="bilinear") + 0.05 * rand((2, 3, 160, 160), 6)
    img = img.clamp(0, 1)
    ud = USMSharp().to("cuda:0")
    blur_d = ud._blur(img.to("cuda:0")).cpu()
    assert float((blur_d - filter2D(img, u.kernel)).abs().max()) <= 5e-6          # (2 601-term fp32 sums in two orders: measured 2.1e-6)
    d = (ud(img.to("cuda:0")).cpu() - u(img)).abs()
    assert float(d.mean()) <= 1e-5 and float((d > 1e-4).float().mean()) <= 2e-3, (float(d.mean()), float(d.max()))


@pytest.mark.gpu
def test_discriminator_convs_on_libsrbh_match_the_fixture_and_the_stock_graph(golden_dir):
    """UNetDiscriminatorSN's 3x3 stride-1 convs (conv0, conv4..conv9, SR/rrdbnet_arch.py:256-265,285-301) on the head's convolution kernels
    (`libsrbh = "f32"`): the reference's CPU output (g14 `disc_out`) within 2e-5; and, at a width whose convs need output / input slicing
    (num_feat 32: conv4 is 256 -> 128), forward and EVERY gradient -- input, spectral-norm `weight_orig`s, biases -- against the stock graph
    in float64 from the same state: exact mode <= 1e-4, 16-bit operand mode output <= 5e-3 and gradient cosines >= 0.99."""
    import copy
    from srbh_amd.srgan import UNetDiscriminatorSN
    g = _g14(golden_dir)
    d = UNetDiscriminatorSN(3, num_feat=8, skip_connection=True).eval()
    d.load_state_dict({k[len("disc_sd_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("disc_sd_")}, strict=True)
    d = d.to("cuda:0")
    d.libsrbh = "f32"
    with torch.no_grad():
        y = d(rand((2, 3, 32, 32), 143, 0.0, 1.0).to("cuda:0"))
    assert O.rel_l2(y.cpu(), torch.from_numpy(g["disc_out"])) <= 2e-5
    torch.manual_seed(5)
    base = UNetDiscriminatorSN(3, num_feat=32, skip_connection=True).train()
    x0 = rand((2, 3, 64, 64), 7, 0.0, 1.0)
    wgt = rand((2, 1, 64, 64), 8)
    res = {}
    # the yardstick is the stock graph in FLOAT64 on the CPU (MIOpen's fp32 convolutions are themselves only ~1e-3 accurate with some of the
    # solvers it picks on a fresh box: a device-stock-vs-libsrbh comparison at 1e-4 failed one run in four there)
    for mode in ("cpu64", "f32", "f16"):
        m = copy.deepcopy(base)
        if mode == "cpu64":
            m = m.double()
            x = x0.double().clone().requires_grad_(True)
            w_ = wgt.double()
        else:
            m = m.to("cuda:0")
            m.libsrbh = mode
            x = x0.to("cuda:0").clone().requires_grad_(True)
            w_ = wgt.to("cuda:0")
        y = m(x)
        (y * w_).sum().backward()
        res[mode] = (y.detach().cpu().float(), x.grad.cpu().float(), {k: p.grad.cpu().float() for k, p in m.named_parameters() if p.grad is not None})
    cos = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()).clamp_min(1e-300))   # noqa: E731
    y0, gx0, g0 = res["cpu64"]
    assert len(g0) == 12
    y1, gx1, g1 = res["f32"]
    assert O.rel_l2(y1, y0) <= 1e-4 and O.rel_l2(gx1, gx0) <= 1e-4
    for k in g0:
        assert O.rel_l2(g1[k], g0[k]) <= 1e-4, (k, O.rel_l2(g1[k], g0[k]))
    y2, gx2, g2 = res["f16"]
    assert O.rel_l2(y2, y0) <= 5e-3 and cos(gx2, gx0) >= 0.99
    for k in g0:
        assert cos(g2[k], g0[k]) >= 0.99, (k, cos(g2[k], g0[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 8, 5, 7), (1, 64, 16, 16), (3, 4, 1, 1)])
def test_bilinear_x2_kernels_equal_interpolate(shape):
    """srbh_bilinear2x_nhwc_f32 (UNetDiscriminatorSN's up-sampling, SR/rrdbnet_arch.py:285-297) and its gather-form adjoint against
    F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) and its autograd, borders and a 1 x 1 plane included: <= 1e-6."""
    from srbh_amd.srgan import _Bilinear2xFn
    x = rand(shape, 21).to("cuda:0")
    w = rand((shape[0], shape[1], 2 * shape[2], 2 * shape[3]), 22).to("cuda:0")
    outs = []
    for fn in (lambda t: torch.nn.functional.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False), _Bilinear2xFn.apply):
        xx = x.clone().requires_grad_(True)
        y = fn(xx)
        (y * w).sum().backward()
        outs.append((y.detach().cpu(), xx.grad.cpu()))
    assert outs[0][0].shape == outs[1][0].shape
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 1e-6 and float((outs[0][1] - outs[1][1]).abs().max()) <= 2e-6
