"""``SRRegress_Cls_feature`` -- the height / building-hierarchy model that train.py:143-148 and
predict_realesanet_feature_globe.py:90-93 instantiate (reference mymodels.py:233-337), on MI355X.

Same constructor kwargs, ``forward`` / ``forward_unsup`` / ``forward_nobuild`` signatures, returned tuples (no
squeeze, callers squeeze: train.py:248-249) and state_dict prefixes (``encoder.``, ``decoder1.``, ``decoder2.``,
``reg.``, ``seg.``, ``hrfeat.``, ``aggre_height.``).  The 256x256 part -- ``hrfeat`` (HRfeature) and ``reg`` / ``seg``
(HRfuse_residual) -- and ``aggre_height`` run on libsrbh kernels with hand-written backward; the < 1 GFLOP/tile
EfficientNet-B4 encoder and the two U-Net decoders use stock PyTorch-ROCm ops.
"""
from __future__ import annotations

import torch
from torch import nn

from . import hrfuse as H
from .encoders import UnetDecoder, get_encoder
from .hrfuse import HRfeature, HRfuse_residual

__all__ = ["SRRegress_Cls_feature"]

import os as _os

# encoder / decoders on a side stream next to the HR head (see _forward_two_streams): "auto" (default) = when no graph is recorded
# (inference: measured +12 % on the tiled-predict path; the training step is bound by the host's op dispatch and gains nothing),
# "1" = always, "0" = never
SIDE_STREAM = _os.environ.get("SRBH_SIDE_STREAM", "auto")
# hrfeat's output as an fp16 tensor inside the inference chain (1, default since round 4) or fp32 (0).  Round 3 had it off: reg / seg read
# it next to the fp32 up-sampler output, and the fused block-entry kernel takes sources of ONE element type.  Since round 4 the
# up-sampler's PixelShuffle store writes fp16 too (hrfuse.Upsampler.forward(out_h16=True)), so both sources of the reg / seg entries
# are fp16: the same numbers (each value is rounded once, where the consumer's staging would round it), ~2 GB less traffic per 128 tiles.
HRFEAT_OUT_H16 = _os.environ.get("SRBH_HRFEAT_OUT_H16", "1") == "1"
# training: issue hrfeat before the encoder (their backward order is then encoder -> hrfeat; see _forward_impl).  0 = upstream's issue order (A/B aid)
HRFEAT_FIRST = _os.environ.get("SRBH_HRFEAT_FIRST", "1") == "1"
DEC2_SIDE = _os.environ.get("SRBH_DEC2_SIDE", "0") == "1"       # pipelined step: decoder2 on its own stream beside decoder1 (A/B aid, see _forward_lr_first)


class SRRegress_Cls_feature(torch.nn.Module):
    def __init__(self, encoder_name="resnet50", encoder_weights="imagenet", encoder_depth=5, in_channels=7, classes=1,
                 super_in=4, super_mid=64, upscale=4, isaggre=False, chans_build=2, uniform_range=0.3, isunsup=False):
        super().__init__()
        # classes / uniform_range / isunsup are accepted and ignored, as upstream (mymodels.py:234-268)
        self.encoder = get_encoder(encoder_name, in_channels=in_channels, depth=encoder_depth, weights=encoder_weights)
        dec_in = (256, 128, 64, 32, 16)
        center = encoder_name.startswith("vgg")
        self.decoder1 = UnetDecoder(encoder_channels=self.encoder.out_channels, decoder_channels=dec_in,
                                    n_blocks=encoder_depth, use_batchnorm=True, center=center, attention_type=None)
        self.decoder2 = UnetDecoder(encoder_channels=self.encoder.out_channels, decoder_channels=dec_in,
                                    n_blocks=encoder_depth, use_batchnorm=True, center=center, attention_type=None)
        self.reg = HRfuse_residual(hr_chans=super_mid, lr_chans=dec_in[-1], mid_chans=dec_in[-1], out_chans=1,
                                   upscale=upscale)
        self.seg = HRfuse_residual(hr_chans=super_mid, lr_chans=dec_in[-1], mid_chans=dec_in[-1], out_chans=chans_build,
                                   upscale=upscale)
        self.hrfeat = HRfeature(in_chans=super_in, mid_chans=super_mid, out_chans=super_mid)
        self.isaggre = isaggre
        if self.isaggre:
            self.aggre_height = nn.Conv2d(super_mid, 1, 3, 1, 1)
            self._plast = H._PackedConv()

    def _aggre(self, height_fea):
        return H._LastConv.run(self, self.aggre_height, height_fea)

    def forward(self, x, super_fea):
        """x: (B,in_channels,64,64) Sentinel-2+1 tile; super_fea: (B,super_in,256,256) RRDBNet.forward_feature output.
        Order of ops as upstream (mymodels.py:270-293)."""
        with H.defer_batch_counters():     # one fused num_batches_tracked increment for all training-mode BatchNorms
            return self._forward_impl(x, super_fea)

    def _forward_impl(self, x, super_fea):
        if callable(super_fea) and not torch.is_tensor(super_fea):
            return self._forward_lr_first(x, super_fea)
        if x.is_cuda and (SIDE_STREAM == "1" or (SIDE_STREAM == "auto" and not torch.is_grad_enabled())):
            return self._forward_two_streams(x, super_fea)
        # Same ops as upstream; only the ISSUE order of the two independent first ops differs under a recorded graph (HRFEAT_FIRST): autograd
        # runs ready nodes newest-first, so with hrfeat recorded FIRST its backward -- 3 ms of chip-filling kernels -- runs LAST, behind the
        # encoder's ~4 ms of 10-30 us kernels, and the next step's persistent trunk kernel starts on a chip whose clocks are up: measured
        # (tools/trunk_gap_probe.py, profiles/r05l_trunk_gap_probe.txt) the two trunk launches take 7.86 ms back to back or behind HBM-bound
        # kernels (8.04) but 9.19 ms behind 5 ms of tiny kernels (8.85 behind an idle gap): the power management has clocked the shader engines down.
        if HRFEAT_FIRST and torch.is_grad_enabled():
            super_fea = self.hrfeat(super_fea, out_h16=HRFEAT_OUT_H16)
            encode_fea = self.encoder(x)
        else:
            encode_fea = self.encoder(x)
            super_fea = self.hrfeat(super_fea, out_h16=HRFEAT_OUT_H16)     # (fp16 NHWC inside the inference chain; reg / seg read it as such)
        height_fea = self.decoder1(*encode_fea)
        if self.isaggre:
            height_aggre = self._aggre(height_fea)
        height = self.reg(height_fea, super_fea)
        build = self.decoder2(*encode_fea)
        build = self.seg(build, super_fea)
        if self.isaggre:
            return height, build, height_aggre
        return height, build

    def _forward_lr_first(self, x, super_fea):
        """Same ops as `forward`, for features that are STILL BEING COMPUTED on another stream (`super_fea` is a handle: calling it makes
        the current stream wait and returns the tensor; harness.TrainStep's pipelined step).  Everything that does not need them is issued
        first -- encoder and both decoders, ~400 small kernels -- so that (a) they run beside the trunk kernel, which leaves CUs free for
        them, and (b) autograd, which runs ready nodes newest-first, runs their backward LAST: the small-kernel chains of one step's end and
        the next step's start form one window for the next batch's trunk."""
        encode_fea = self.encoder(x)
        if DEC2_SIDE and x.is_cuda:
            # the two decoders read the same features and are independent chains of small kernels: the second one on its own stream
            # (autograd replays each op's backward on the stream its forward ran on: their backward overlaps the same way)
            cur = torch.cuda.current_stream(x.device)
            side = self.__dict__.get("_dec2_stream")
            if side is None or side.device != x.device:
                side = self.__dict__["_dec2_stream"] = torch.cuda.Stream(device=x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                build = self.decoder2(*encode_fea)
            for t in encode_fea:
                if torch.is_tensor(t):
                    t.record_stream(side)
            height_fea = self.decoder1(*encode_fea)
            height_aggre = self._aggre(height_fea) if self.isaggre else None
            cur.wait_stream(side)
            build.record_stream(cur)
        else:
            height_fea = self.decoder1(*encode_fea)
            height_aggre = self._aggre(height_fea) if self.isaggre else None
            build = self.decoder2(*encode_fea)
        super_fea = self.hrfeat(super_fea(), out_h16=HRFEAT_OUT_H16)
        height = self.reg(height_fea, super_fea)
        build = self.seg(build, super_fea)
        if self.isaggre:
            return height, build, height_aggre
        return height, build

    # the three independent-until-the-end parts of forward (mymodels.py:270-293): harness._PredictGraph records each into its own HIP
    # graph and replays the first on a second stream (a forked capture inside ONE graph does not run its branches concurrently on
    # ROCm 7.2: tools/graph_branch_probe.py)
    def forward_lr(self, x):
        """encoder + both U-Net decoders on the 64x64 tile: (height_fea, build_fea, height_aggre | None)"""
        encode_fea = self.encoder(x)
        height_fea = self.decoder1(*encode_fea)
        build_fea = self.decoder2(*encode_fea)
        return height_fea, build_fea, (self._aggre(height_fea) if self.isaggre else None)

    def forward_hr(self, super_fea):
        """the RRDBNet features through HRfeature (fp16 NHWC inside the inference chain; reg / seg read it as such)"""
        return self.hrfeat(super_fea, out_h16=HRFEAT_OUT_H16)

    def forward_fuse(self, height_fea, build_fea, super_fea):
        """reg / seg on forward_lr's and forward_hr's results: (height, build)"""
        return self.reg(height_fea, super_fea), self.seg(build_fea, super_fea)

    def _forward_two_streams(self, x, super_fea):
        """Same ops, two HIP streams: the EfficientNet encoder and the two U-Net decoders are ~500 small stock-op launches at
        64x64 and below (dispatch-latency bound, a few CUs each), the 256x256 HR head is a handful of chip-filling kernels;
        they do not depend on each other until `reg` / `seg` need the decoder outputs.  Autograd replays each op's backward on the
        stream its forward ran on, so the backward overlaps the same way."""
        cur = torch.cuda.current_stream(x.device)
        side = self.__dict__.get("_side_stream")
        if side is None or side.device != x.device:
            side = torch.cuda.Stream(device=x.device)
            self.__dict__["_side_stream"] = side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            height_fea, build_fea, height_aggre = self.forward_lr(x)
        super_fea = self.forward_hr(super_fea)
        cur.wait_stream(side)
        for t in (height_fea, build_fea, height_aggre):      # produced on `side`, consumed on `cur`: keep the allocator from recycling early
            if t is not None:
                t.record_stream(cur)
        height, build = self.forward_fuse(height_fea, build_fea, super_fea)
        if self.isaggre:
            return height, build, height_aggre
        return height, build

    def forward_unsup(self, x, super_fea):
        """mymodels.py:295-313: height only, squeezed."""
        encode_fea = self.encoder(x)
        super_fea = self.hrfeat(super_fea)
        height_fea = self.decoder1(*encode_fea)
        return self.reg(height_fea, super_fea).squeeze()

    def forward_nobuild(self, x, super_fea):
        """mymodels.py:315-337: skips decoder2 / seg."""
        encode_fea = self.encoder(x)
        super_fea = self.hrfeat(super_fea, out_h16=HRFEAT_OUT_H16)     # (fp16 NHWC inside the inference chain; reg / seg read it as such)
        height_fea = self.decoder1(*encode_fea)
        if self.isaggre:
            height_aggre = self._aggre(height_fea)
        height = self.reg(height_fea, super_fea)
        if self.isaggre:
            return height, height_aggre
        return height
