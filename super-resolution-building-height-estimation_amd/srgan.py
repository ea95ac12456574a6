"""SR-stage (GAN fine-tuning) companions of RRDBNet, SURVEY.md 8f-4: the pieces around the generator that the reference's
``RealESRGAN`` trainer uses (SR/rrdbnet_arch.py:244-303 UNetDiscriminatorSN, :387-434 filter2D / USMSharp; SR/srloss.py:144-249
GANLoss).  They are NOT on the MI355X hot path and stay on stock PyTorch-ROCm ops (VERDICT r01 item 7 allows that); the
generator's forward AND backward run on libsrbh (rrdbnet_autograd.py).  The VGG19 perceptual loss needs torchvision's
pretrained network (absent offline): it is an optional plug-in (``RealESRGAN.cri_perceptual``), not restated."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.utils import spectral_norm

__all__ = ["UNetDiscriminatorSN", "filter2D", "USMSharp", "GANLoss"]


# ---- the discriminator's 3x3 stride-1 convolutions on libsrbh (round 5) --------------------------------------------------------------------
# conv0, conv4 .. conv9 of UNetDiscriminatorSN (SR/rrdbnet_arch.py:256-265,285-301) are 3x3 / stride 1 / padding 1: 74 % of the network's FLOPs.
# They run on the head's convolution kernels (csrc/srbh_head.hip, srbh_head_bwd.hip: forward, data gradient = the forward kernel on transposed +
# flipped weights, weight gradient as a GEMM over pixels) in slices of <= 64 output channels (the kernels' limit), in the precision of the SR-stage
# mode: exact fp32, or fp16 forward / bf16 gradient operands with fp32 accumulation (`precision`).  The three 4x4 stride-2 convs, the bilinear
# up-sampling, LeakyReLU and the skip additions stay stock device ops.  The spectral-norm reparametrisation is torch's own: the hook computes
# `weight = weight_orig / sigma` (with its power iteration in training), the convolution takes that tensor and returns its gradient.
# UNetDiscriminatorSN.libsrbh selects it; the trainer turns it on with the generator's 16-bit operand modes (RealESRGAN.optimize_parameters: 57.5 ->
# 52.2 ms per iteration at batch 8, profiles/r05cs).  Restricting it to the <= 64-channel convs (one launch each) gained nothing.
def _disc_pack(w, cout, cin, transpose, h16, bf16):
    from . import _lib
    L = _lib.lib()
    if h16:
        buf = torch.empty(L.srbh_hpack_h16_bytes(cout, cin, 3) // 2, dtype=torch.float16, device=w.device)
        _lib.check(L.srbh_hpack_conv_h16(w.data_ptr(), cout, cin, 3, int(transpose), int(bf16), buf.data_ptr(), _lib.stream_ptr()), "hpack_conv_h16(disc)")
    else:
        buf = torch.empty(L.srbh_hpack_bytes(cout, cin, 3) // 4, dtype=torch.float32, device=w.device)
        _lib.check(L.srbh_hpack_conv_f32(w.data_ptr(), cout, cin, 3, int(transpose), buf.data_ptr(), _lib.stream_ptr()), "hpack_conv_f32(disc)")
    return buf


def _disc_launch(xn, pack, bias, cout, h16, bf16):
    """one srbh_hconv call: xn NHWC fp32 (B, c0, H, W) -> (B, cout <= 64, H, W) NHWC fp32"""
    import ctypes as C
    from . import _lib
    from . import hrfuse as H
    B, c0, Hh, Ww = xn.shape
    a = _lib.HConvArgs()
    a.src0, a.c0 = xn.data_ptr(), c0
    a.w = pack.data_ptr()
    a.bias = None if bias is None else bias.data_ptr()
    a.cout, a.ksize = cout, 3
    a.B, a.H, a.W = B, Hh, Ww
    out = H.empty_nhwc(B, cout, Hh, Ww, xn.device)
    a.out = out.data_ptr()
    L = _lib.lib()
    if h16:
        _lib.check(L.srbh_hconv_h16(C.byref(a), int(bf16), _lib.stream_ptr()), "hconv_h16(disc)")
    else:
        _lib.check(L.srbh_hconv_f32(C.byref(a), _lib.stream_ptr()), "hconv_f32(disc)")
    return out


class _DiscConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, h16):
        from . import hrfuse as H
        xn = H.to_nhwc(x.detach().float())
        w = weight.detach().float().contiguous()
        cout, cin = w.shape[:2]
        outs = []
        for lo in range(0, cout, 64):
            n = min(64, cout - lo)
            b = None
            if bias is not None:
                b = torch.zeros((n + 15) // 16 * 16, dtype=torch.float32, device=x.device)
                b[:n] = bias.detach().float()[lo:lo + n]
            outs.append(_disc_launch(xn, _disc_pack(w[lo:lo + n].contiguous(), n, cin, False, h16, False), b, n, h16, False))
        ctx.save_for_backward(xn, w)
        ctx.h16, ctx.has_bias = bool(h16), bias is not None
        return outs[0] if len(outs) == 1 else H.to_nhwc(torch.cat(outs, dim=1))

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        from . import _lib
        from . import hrfuse as H
        xn, w = ctx.saved_tensors
        h16 = ctx.h16
        gn = H.to_nhwc(g.float())
        cout, cin = w.shape[:2]
        L = _lib.lib()
        dx = None
        if ctx.needs_input_grad[0]:          # conv^T: the forward kernel on transposed + flipped weight slices, <= 64 input channels per launch
            parts = []
            for lo in range(0, cin, 64):
                n = min(64, cin - lo)
                parts.append(_disc_launch(gn, _disc_pack(w[:, lo:lo + n].contiguous(), n, cout, True, h16, True), None, n, h16, True))
            dx = parts[0] if len(parts) == 1 else H.to_nhwc(torch.cat(parts, dim=1))
        dw = None
        if ctx.needs_input_grad[1]:
            dws = []
            for lo in range(0, cout, 64):
                n = min(64, cout - lo)
                gs = gn if cout <= 64 else H.to_nhwc(gn[:, lo:lo + n].contiguous(memory_format=torch.channels_last))
                d = torch.empty((n, cin, 3, 3), dtype=torch.float32, device=g.device)
                a = _lib.HWGradArgs()
                a.src0, a.c0 = xn.data_ptr(), cin
                a.dy, a.cout, a.ksize = gs.data_ptr(), n, 3
                a.B, a.H, a.W = xn.shape[0], xn.shape[2], xn.shape[3]
                a.dw = d.data_ptr()
                ws = torch.empty(L.srbh_hwgrad_ws_bytes(n, cin, 3) // 4, dtype=torch.float32, device=g.device)
                a.ws = ws.data_ptr()
                _lib.check((L.srbh_hconv_wgrad_b16 if h16 else L.srbh_hconv_wgrad_f32)(C.byref(a), _lib.stream_ptr()), "hconv_wgrad(disc)")
                dws.append(d)
            dw = dws[0] if len(dws) == 1 else torch.cat(dws, dim=0)
        db = gn.sum(dim=(0, 2, 3)) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


def _disc_conv3x3(conv, x, h16):
    """conv(x) of a 3x3 stride-1 discriminator conv on libsrbh.  torch's spectral_norm is a forward PRE-hook that sets conv.weight =
    weight_orig / sigma (one power iteration in training): it is the only hook this path knows how to honour, so it is called by hand and a
    module carrying ANY other hook (forward hooks, foreign pre-hooks, hooks with kwargs) takes the stock `conv(x)` with the full hook protocol."""
    from torch.nn.utils.spectral_norm import SpectralNorm
    pre = list(conv._forward_pre_hooks.values())
    if (any(not isinstance(h, SpectralNorm) for h in pre) or conv._forward_hooks or conv._backward_hooks or conv._backward_pre_hooks
            or getattr(conv, "_forward_pre_hooks_with_kwargs", None) or getattr(conv, "_forward_hooks_with_kwargs", None)):
        return conv(x)
    for hook in pre:
        hook(conv, (x,))
    return _DiscConvFn.apply(x, conv.weight, conv.bias, h16)


class _Bilinear2xFn(torch.autograd.Function):
    """F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) on a device tensor through srbh_bilinear2x_nhwc_f32 (the stock backward
    scatters with atomics: 4.5 ms of a 52 ms trainer iteration at batch 8)"""

    @staticmethod
    def forward(ctx, x):
        from . import _lib
        from . import hrfuse as H
        xn = H.to_nhwc(x.detach().float())
        B, Cc, Hh, Ww = xn.shape
        out = H.empty_nhwc(B, Cc, 2 * Hh, 2 * Ww, x.device)
        _lib.check(_lib.lib().srbh_bilinear2x_nhwc_f32(xn.data_ptr(), out.data_ptr(), B, Hh, Ww, Cc, 0, _lib.stream_ptr()), "bilinear2x(fwd)")
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        from . import hrfuse as H
        gn = H.to_nhwc(g.float())
        B, Cc, H2, W2 = gn.shape
        dx = H.empty_nhwc(B, Cc, H2 // 2, W2 // 2, g.device)
        _lib.check(_lib.lib().srbh_bilinear2x_nhwc_f32(gn.data_ptr(), dx.data_ptr(), B, H2 // 2, W2 // 2, Cc, 1, _lib.stream_ptr()), "bilinear2x(bwd)")
        return dx


class UNetDiscriminatorSN(nn.Module):
    """U-Net discriminator with spectral normalisation; state_dict keys conv0..conv9 (conv1..conv8 carry weight_orig / weight_u /
    weight_v of torch's spectral_norm) as upstream (SR/rrdbnet_arch.py:244-303)."""

    def __init__(self, num_in_ch, num_feat=64, skip_connection=True):
        super().__init__()
        self.skip_connection = skip_connection
        nf = num_feat
        self.conv0 = nn.Conv2d(num_in_ch, nf, 3, 1, 1)
        self.conv1 = spectral_norm(nn.Conv2d(nf, nf * 2, 4, 2, 1, bias=False))
        self.conv2 = spectral_norm(nn.Conv2d(nf * 2, nf * 4, 4, 2, 1, bias=False))
        self.conv3 = spectral_norm(nn.Conv2d(nf * 4, nf * 8, 4, 2, 1, bias=False))
        self.conv4 = spectral_norm(nn.Conv2d(nf * 8, nf * 4, 3, 1, 1, bias=False))
        self.conv5 = spectral_norm(nn.Conv2d(nf * 4, nf * 2, 3, 1, 1, bias=False))
        self.conv6 = spectral_norm(nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=False))
        self.conv7 = spectral_norm(nn.Conv2d(nf, nf, 3, 1, 1, bias=False))
        self.conv8 = spectral_norm(nn.Conv2d(nf, nf, 3, 1, 1, bias=False))
        self.conv9 = nn.Conv2d(nf, 1, 3, 1, 1)

    libsrbh = None          # None: stock convolutions; "f32" / "f16": the 3x3 stride-1 convs on libsrbh in that operand precision (device tensors only)

    def forward(self, x):
        act = lambda t: F.leaky_relu(t, 0.2)
        up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
        if self.libsrbh is not None and x.is_cuda:
            h16 = self.libsrbh == "f16"
            c3 = lambda conv, t: _disc_conv3x3(conv, t, h16)      # noqa: E731
            up = lambda t: _Bilinear2xFn.apply(t) if t.shape[1] % 4 == 0 else F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)      # noqa: E731
        else:
            c3 = lambda conv, t: conv(t)                           # noqa: E731
        s2 = lambda conv, t: conv(t)                               # noqa: E731  (4x4 stride 2: stock convolutions)
        x0 = act(c3(self.conv0, x))
        x1 = act(s2(self.conv1, x0))
        x2 = act(s2(self.conv2, x1))
        x3 = act(s2(self.conv3, x2))
        x4 = act(c3(self.conv4, up(x3)))
        if self.skip_connection:
            x4 = x4 + x2
        x5 = act(c3(self.conv5, up(x4)))
        if self.skip_connection:
            x5 = x5 + x1
        x6 = act(c3(self.conv6, up(x5)))
        if self.skip_connection:
            x6 = x6 + x0
        return c3(self.conv9, act(c3(self.conv8, act(c3(self.conv7, x6)))))


def filter2D(img, kernel):
    """cv2.filter2D on (b,c,h,w) with a (1|b, k, k) kernel, reflect padding (SR/rrdbnet_arch.py:387-409)."""
    k = kernel.size(-1)
    if k % 2 != 1:
        raise ValueError("Wrong kernel size")
    b, c, h, w = img.shape
    img = F.pad(img, (k // 2,) * 4, mode="reflect")
    ph, pw = img.shape[-2:]
    if kernel.size(0) == 1:
        return F.conv2d(img.reshape(b * c, 1, ph, pw), kernel.view(1, 1, k, k)).view(b, c, h, w)
    kern = kernel.view(b, 1, k, k).repeat(1, c, 1, 1).view(b * c, 1, k, k)
    return F.conv2d(img.reshape(1, b * c, ph, pw), kern, groups=b * c).view(b, c, h, w)


def _gaussian_kernel_1d(ksize, sigma):
    """cv2.getGaussianKernel: sigma <= 0 -> 0.3*((ksize-1)*0.5 - 1) + 0.8; exp(-(i-c)^2 / (2 sigma^2)) normalised to sum 1."""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    c = (ksize - 1) * 0.5
    g = torch.tensor([math.exp(-((i - c) ** 2) / (2.0 * sigma * sigma)) for i in range(ksize)], dtype=torch.float64)
    return (g / g.sum()).float()


class USMSharp(nn.Module):
    """unsharp masking (SR/rrdbnet_arch.py:412-434): blur with a radius x radius Gaussian, sharpen where |residual|*255 > threshold."""

    def __init__(self, radius=50, sigma=0):
        super().__init__()
        if radius % 2 == 0:
            radius += 1
        self.radius = radius
        g = _gaussian_kernel_1d(radius, sigma)
        self.register_buffer("kernel", torch.outer(g, g).unsqueeze(0))
        self.register_buffer("kernel1d", g.clone(), persistent=False)      # (not in the state_dict: the reference's has `kernel` only)

    def _blur(self, img):
        """filter2D(img, self.kernel).  On a device tensor the Gaussian is applied as its two 1-D factors (the kernel IS their outer product):
        2 x 51 taps instead of 2 601 per pixel -- the 2-D form cost 7 ms per feed_data at batch 8 (tools/sr_iteration_phases.py); same values up
        to fp32 summation order.  CPU tensors keep the 2-D form the reference fixture is compared against."""
        if not img.is_cuda:
            return filter2D(img, self.kernel)
        k = self.kernel1d.numel()
        b, c, h, w = img.shape
        x = F.pad(img, (k // 2,) * 4, mode="reflect").reshape(b * c, 1, h + k - 1, w + k - 1)
        x = F.conv2d(x, self.kernel1d.view(1, 1, 1, k))
        return F.conv2d(x, self.kernel1d.view(1, 1, k, 1)).view(b, c, h, w)

    def forward(self, img, weight=0.5, threshold=10):
        blur = self._blur(img)
        residual = img - blur
        soft_mask = self._blur((residual.abs() * 255 > threshold).float())
        sharp = torch.clip(img + weight * residual, 0, 1)
        return soft_mask * sharp + (1 - soft_mask) * img


# ---- VGG19 perceptual loss (SR/srloss.py:61-143; wired at SR/rrdbnet_arch.py:496-498, used at :559-562) -------------------------------
# The reference builds `torchvision.models.vgg19(weights="IMAGENET1K_V1").features` and downloads its weights.  Neither torchvision nor a
# network exists offline, so the feature stack is restated from the layer table the reference itself prints (SR/srloss.py:8-48: configuration
# "E": 2 x 64, pool, 2 x 128, pool, 4 x 256, pool, 4 x 512, pool, 4 x 512, pool; every conv 3x3 / pad 1 + ReLU, max-pool 2 / 2) and takes the
# weights as a state_dict with torchvision's keys (`features.<i>.weight` / `.bias`, or the bare `<i>.weight` of the `features` module).
# Without one the convolutions keep their random initialisation -- a structure test, not a perceptual metric; RealESRGAN(is_train=True) only
# adds the term when it is handed weights or a ready module.
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


def vgg19_features():
    """torchvision.models.vgg19().features: indices 0..36 as listed in SR/srloss.py:8-48"""
    layers, cin = [], 3
    for v in VGG19_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


class VGGFeatureExtractor(nn.Module):
    """SR/srloss.py:61-105: the VGG19 feature stack cut behind `feature_layer` (a list: one nn.Sequential child per cut, keys
    `features.child<i>.<j>.*` as upstream; an int: a single stack), optional ImageNet input normalisation / [-1, 1] -> [0, 1] range map,
    parameters frozen.  state_dict: torchvision's vgg19 weights (see above)."""

    def __init__(self, feature_layer=(2, 7, 16, 25, 34), use_input_norm=True, use_range_norm=False, state_dict=None):
        super().__init__()
        feats = vgg19_features()
        if state_dict is not None:
            sd = {(k[len("features."):] if k.startswith("features.") else k): v for k, v in state_dict.items()
                  if k.startswith("features.") or k.split(".")[0].isdigit()}
            feats.load_state_dict(sd, strict=True)
        self.use_input_norm = use_input_norm
        self.use_range_norm = use_range_norm
        if self.use_input_norm:
            self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
            self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
        self.list_outputs = isinstance(feature_layer, (list, tuple))
        children = list(feats.children())
        if self.list_outputs:
            self.features = nn.Sequential()
            cuts = [-1] + list(feature_layer)
            for i in range(len(cuts) - 1):
                self.features.add_module("child" + str(i), nn.Sequential(*children[cuts[i] + 1:cuts[i + 1] + 1]))
        else:
            self.features = nn.Sequential(*children[:feature_layer + 1])
        for v in self.features.parameters():
            v.requires_grad = False

    def forward(self, x):
        if self.use_range_norm:
            x = (x + 1.0) / 2.0
        if self.use_input_norm:
            x = (x - self.mean) / self.std
        if self.list_outputs:
            out = []
            for child in self.features.children():
                x = child(x)
                out.append(x.clone())      # (the next child starts with an in-place ReLU)
            return out
        return self.features(x)


class PerceptualLoss(nn.Module):
    """SR/srloss.py:108-143: sum_i weights[i] * L1|MSE(vgg_i(x), vgg_i(gt.detach())) -- times `self.loss_weight`, which upstream pins to 1.0
    whatever the argument says (srloss.py:123: kept, it is the arithmetic the checkpoints were trained with)."""

    def __init__(self, feature_layer=(2, 7, 16, 25, 34), weights=(0.1, 0.1, 1.0, 1.0, 1.0), lossfn_type="l1", use_input_norm=True,
                 use_range_norm=False, loss_weight=1.0, state_dict=None):
        super().__init__()
        self.vgg = VGGFeatureExtractor(feature_layer=feature_layer, use_input_norm=use_input_norm, use_range_norm=use_range_norm,
                                       state_dict=state_dict)
        self.lossfn_type = lossfn_type
        self.weights = list(weights)
        self.lossfn = nn.L1Loss() if lossfn_type == "l1" else nn.MSELoss()
        self.loss_weight = 1.0

    def forward(self, x, gt):
        x_vgg, gt_vgg = self.vgg(x), self.vgg(gt.detach())
        loss = 0.0
        if isinstance(x_vgg, list):
            for i in range(len(x_vgg)):
                loss = loss + self.weights[i] * self.lossfn(x_vgg[i], gt_vgg[i])
        else:
            loss = loss + self.lossfn(x_vgg, gt_vgg.detach())
        return loss * self.loss_weight


class GANLoss(nn.Module):
    """SR/srloss.py:144-249: 'vanilla' (BCE with logits), 'lsgan', 'wgan', 'wgan_softplus', 'hinge'; loss_weight applies to the
    generator only."""

    def __init__(self, gan_type, real_label_val=1.0, fake_label_val=0.0, loss_weight=1.0):
        super().__init__()
        if gan_type not in ("vanilla", "lsgan", "wgan", "wgan_softplus", "hinge"):
            raise NotImplementedError(f"GAN type {gan_type} is not implemented.")
        self.gan_type, self.loss_weight = gan_type, loss_weight
        self.real_label_val, self.fake_label_val = real_label_val, fake_label_val

    def forward(self, input, target_is_real, is_disc=False):
        t = self.gan_type
        if t == "hinge":
            if is_disc:
                loss = F.relu(1 + (-input if target_is_real else input)).mean()
            else:
                loss = -input.mean()
        elif t == "wgan":
            loss = -input.mean() if target_is_real else input.mean()
        elif t == "wgan_softplus":
            loss = F.softplus(-input).mean() if target_is_real else F.softplus(input).mean()
        else:
            target = input.new_full(input.shape, self.real_label_val if target_is_real else self.fake_label_val)
            loss = F.binary_cross_entropy_with_logits(input, target) if t == "vanilla" else F.mse_loss(input, target)
        return loss if is_disc else loss * self.loss_weight
