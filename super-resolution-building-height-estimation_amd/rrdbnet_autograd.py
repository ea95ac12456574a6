"""Training path of RRDBNet: forward AND backward of the whole generator on libsrbh kernels (first slice).

The height stage never differentiates the RRDB stack (train.py:139-140,243-244); the SR stage does
(reference SR/rrdbnet_arch.py:538-592: ``l_g_total.backward()`` through ``self.net_g``).  This module is the autograd
``Function`` behind ``RRDBNet.forward`` / ``forward_feature`` when the training path is switched on
(``RRDBNet.enable_training_path()``; ``RealESRGAN(is_train=True)`` does it):

* forward  = the strict exact-fp32 network of ``RRDBNet._run_strict`` (v_mfma_f32_16x16x4_f32 convolutions with the
  LeakyReLU / ``x5*0.2+x`` / ``out*0.2+x`` epilogues fused), keeping every RDB's 192-channel dense buffer
  ``[x | x1 | x2 | x3 | x4]`` (post-activation, fp32 NHWC) -- 3 MiB per tile and RDB, i.e. 13 GB for the 23-block net at
  batch 64: nothing against 288 GB of HBM, so nothing is recomputed;
* backward = per conv: weight gradient as a GEMM over pixels (``srbh_hconv_wgrad_f32`` reading a strided view of the dense
  buffer), bias gradient, data gradient with the transposed + flipped weight through the same forward kernel, accumulated
  IN PLACE into the 192-channel gradient buffer by the conv's residual epilogue (out = conv + res1 on the same view); the
  LeakyReLU mask comes from the saved post-activation plane (x > 0 ? 1 : 0.2, as torch's leaky_relu backward).

Everything is exact fp32 (parity <= 1e-5 against the reference's autograd, fixture g14): the first slice trades speed
(~20x the fp16-MFMA inference path) for having the SR-stage generator step at all; an fp16/bf16-operand backward on the
trunk's MFMA kernel is the follow-up.  Element-wise glue (masks, 2x2 sums of the nearest-x2 backward, bias sums) uses torch
ops on the device.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, wcache
from . import hrfuse as H

__all__ = ["rrdbnet_apply", "set_train_precision"]

# Operand precision of the RRDBNet TRAINING graph: "f32" (default) = exact fp32 matrix cores in forward, data and weight gradients
# (what the <= 2e-3 / float64 parity tests pin); "mixed" = forward convs with fp16 operands, data and weight gradients with bf16
# operands, fp32 accumulation and fp32 residual / LeakyReLU epilogues everywhere (the head's TrainStep policy, hrfuse.py): the
# 16-bit matrix cores run these 64..192-channel convs several times faster.  SRBH_SR_TRAIN_PRECISION / set_train_precision().
import os as _os

_PRECISION = {"mode": _os.environ.get("SRBH_SR_TRAIN_PRECISION", "f32")}


def set_train_precision(mode):
    if mode not in ("f32", "mixed", "fast"):
        raise ValueError("RRDBNet training precision must be 'f32', 'mixed' or 'fast'")
    _PRECISION["mode"] = mode


def _mixed():
    return _PRECISION["mode"] in ("mixed", "fast")


def _fast():
    return _PRECISION["mode"] == "fast"


# ---------------------------------------------------------------------------------------------------------------------------------
# "fast" (round 3, SURVEY 8f-4 second slice): the 69 dense blocks -- 92 % of the network's FLOPs -- forward AND backward on the trunk's
# own kernel family (csrc/srbh_conv3x3_kernel.h: v_mfma_f32_32x32x16, ACT16 chunk planes, LDS-DMA staging), the rest of the network
# (conv_first, conv_body, the up-sampler, conv_hr / conv_last) as in "mixed".
#   forward  : the per-layer launch sequence of the inference path (csrc/srbh_rrdbnet.hip) driven from here, with ONE dense ACT16 buffer
#              PER RDB kept for the backward (fp16 [x | x1 | x2 | x3 | x4]: 1.6 MiB per tile and RDB; the LeakyReLU masks are the signs of
#              the saved planes);
#   backward : the gradient of a dense block is itself a dense block run in reverse: with G = [g5 | g4 | g3 | g2 | g1] (the gradients of
#              the five convs' outputs, g5 = 0.2 g) the gradient of plane X_m is ONE 3x3 conv over the first channels of G with the
#              transposed + flipped weight slices of every conv that consumed X_m stacked along K -- the forward's shapes exactly (64,
#              96, 128, 160 -> 32 and 192 -> 64 channels), so srbh_conv3x3_x16 runs them: bf16 operands (gradients need fp32's exponent
#              range), the LeakyReLU derivative from the saved plane folded into the epilogue, the identity path added as `skip`;
#              weight gradients as GEMMs over pixels straight from the ACT16 planes (srbh_act16_wgrad_b16), bias gradients as plane sums.
_FAST_WS = {}          # geometry -> list of buffer sets; a set is LEASED by one forward until its backward (or its graph's death)


class _FastLease:
    """One forward's claim on a buffer set: the dense planes saved for backward (ws['D']: activations AND the LeakyReLU masks) must
    not be overwritten by another forward at the same geometry before this forward's backward has read them (round-3 ADVICE: two
    generator passes, a second trainable net, a recompute interleaved with another forward -- silently wrong gradients).  Released
    by backward, or when autograd drops the graph without running it."""

    def __init__(self, ws):
        self.ws = ws
        ws["busy"] = True
        ws["gen"] += 1
        self.gen = ws["gen"]

    def check(self):
        if self.ws is None:
            raise RuntimeError("RRDBNet fast training mode: this forward's saved activation planes were released by its first backward "
                               "(a second backward through the same graph, e.g. retain_graph=True, needs a new forward; the exact 'f32' "
                               "training mode keeps its tensors and supports it)")
        if self.ws["gen"] != self.gen or not self.ws["busy"]:
            raise RuntimeError("RRDBNet fast training mode: the saved activation planes of this forward were handed to another forward "
                               "before its backward ran (buffer generation %d, expected %d)" % (self.ws["gen"], self.gen))

    def release(self):
        if self.ws is not None and self.ws["gen"] == self.gen:
            self.ws["busy"] = False
        self.ws = None

    def __del__(self):
        self.release()


def _fast_buffers(B, Hh, Ww, n_rdb, dev):
    """zero-bordered ACT16 buffers (the kernels never write the borders): n_rdb + 1 dense activation buffers, one gradient buffer.
    Returns a set no live forward holds: the cached one when it is free, else a fresh allocation (kept for the next overlap)."""
    key = (B, Hh, Ww, n_rdb, str(dev))
    pool = _FAST_WS.get(key)
    if pool is None:
        _FAST_WS.clear()             # one geometry at a time: a B=24 set is 7 GB
        pool = _FAST_WS[key] = []
    for ws in pool:
        if not ws["busy"]:
            return ws
    L = _lib.lib()
    nb = L.srbh_act16_bytes(B, 192, Hh, Ww)
    nb = (nb + 255) // 256 * 256
    na = L.srbh_rrdbnet_trunk_train_aux_bytes(B, Hh, Ww)          # scratch of the persistent forward / backward (0: this geometry runs the per-layer sequences)
    # G: the gradient planes -- a row of n_rdb + 1 buffers for the persistent backward (every RDB's G is kept for the weight gradients that follow
    # the launch), two (double buffered against the side stream) for the per-layer sequence
    ws = {"nb": nb, "D": torch.zeros((n_rdb + 1) * nb, dtype=torch.uint8, device=dev),
          "G": torch.zeros(((n_rdb + 1) if na and BWD_PERSISTENT else 2) * nb, dtype=torch.uint8, device=dev),
          "wg": torch.empty(2 * (L.srbh_rrdbnet_trunk_wgrad_ws_bytes() // 4), dtype=torch.float32, device=dev), "busy": False, "gen": 0}
    ws["aux"] = torch.zeros(na, dtype=torch.uint8, device=dev) if na else None
    ws["zero_bias"] = torch.zeros(64, dtype=torch.float32, device=dev)
    nt = L.srbh_trunk_wgrad_ws_bytes(n_rdb // 3, B, Hh, Ww) if na and BWD_PERSISTENT else 0      # the one-launch weight gradients behind the persistent backward
    ws["twg"] = torch.empty(nt, dtype=torch.uint8, device=dev) if nt else None
    pool.append(ws)
    return ws


class _TrunkBwdPacks:
    """bf16 WPACK16 images of the five gradient convs of every RDB (stacked transposed + flipped weight slices, see above)"""

    def __init__(self):
        self.key = None

    def get(self, net):
        ps = [p for blk in net.body for r in (1, 2, 3) for k in range(1, 6) for p in (getattr(getattr(blk, f"rdb{r}"), f"conv{k}").weight,)]
        key = tuple((p._version, getattr(p, "_srbh_gen", 0), p.data_ptr()) for p in ps) + wcache.gen()
        if key != self.key:
            L = _lib.lib()
            st = _lib.stream_ptr()
            n = len(ps) // 5
            dev = ps[0].device
            plan = getattr(self, "plan", None)
            with torch.no_grad():
                W = [torch.stack([ps[i * 5 + k].detach().float() for i in range(n)]) for k in range(5)]      # W[k]: (n, cout_k, cin_k, 3, 3)
                tf = lambda w, lo, hi: w[:, :, lo:hi].permute(0, 2, 1, 3, 4).flip(3, 4)                        # noqa: E731  (n, hi-lo, cout_k, 3, 3)
                # plane X_m (dense channels lo:hi) is consumed by conv_{m+1}..conv5; G order: g5, g4, g3, g2, g1
                planes = [(160, 192, [4]), (128, 160, [4, 3]), (96, 128, [4, 3, 2]), (64, 96, [4, 3, 2, 1]), (0, 64, [4, 3, 2, 1, 0])]
                if plan is None or plan["n"] != n or plan["dev"] != dev:
                    # persistent stacked weights + packs + the job table of ONE srbh_pack_conv3x3_many launch (the weights move every iteration:
                    # 345 pack launches and a stream synchronisation per backward otherwise); rewritten in place, so the persistent backward's
                    # cached layer table -- pointers into `buf` -- stays valid
                    import numpy as np
                    shapes = [(n, hi - lo, sum((32, 32, 32, 32, 64)[k] for k in ks), 3, 3) for lo, hi, ks in planes]
                    stk = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
                    sizes = [L.srbh_wpack16_bytes(sh[1], sh[2]) for sh in shapes]
                    offs, tot = [], 0
                    for sz in sizes:
                        offs.append(tot)
                        tot += (sz + 255) // 256 * 256
                    buf = torch.zeros(n * tot, dtype=torch.uint8, device=dev)
                    desc_t = np.dtype([("w", np.uint64), ("packed", np.uint64), ("bias_src", np.uint64), ("bias_dst", np.uint64), ("cout", np.int32),
                                       ("cin", np.int32), ("bf16", np.int32), ("pad", np.int32)])
                    tab = np.zeros(n * 5, dtype=desc_t)
                    for i in range(n):
                        for jj, sh in enumerate(shapes):
                            tab[i * 5 + jj] = (stk[jj][i].data_ptr(), buf.data_ptr() + i * tot + offs[jj], 0, 0, sh[1], sh[2], 1, 0)
                    plan = self.plan = {"n": n, "dev": dev, "stk": stk, "buf": buf, "offs": offs, "tot": tot, "max_elems": max(sizes) // 2,
                                        "table": torch.from_numpy(tab.view(np.uint8).copy()).to(dev)}
                for jj, (lo, hi, ks) in enumerate(planes):
                    torch.cat([tf(W[k], lo, hi) for k in ks], dim=2, out=plan["stk"][jj])
            _lib.check(L.srbh_pack_conv3x3_many(plan["table"].data_ptr(), n * 5, plan["max_elems"], st), "pack_conv3x3_many(bf16)")
            self.key, self.buf, self.offs, self.stride = key, plan["buf"], plan["offs"], plan["tot"]
        wcache.keep(self.buf)
        return self

    def ptr(self, i, j):
        return self.buf.data_ptr() + i * self.stride + self.offs[j]


def _conv16(a_in, in_chunks, w, bias, cout, B, Hh, Ww, *, lrelu=0, out16=None, out16_chunk0=0, res1=None, res2=None, skip=None, out32=None,
            bf16=0, mask=None, mask_chunk0=0):
    a = _lib.ConvArgs()
    a.in_, a.in_chunks_total, a.in_chunk0, a.in_chunks = a_in, 6, 0, in_chunks
    a.w, a.bias, a.cout = w, bias, cout
    a.B, a.H, a.W, a.lrelu = B, Hh, Ww, lrelu
    if res1 is not None:
        a.res_scale, a.res1, a.res1_update = 0.2, res1, 1
    if res2 is not None:
        a.res2_scale, a.res2, a.res2_update = 0.2, res2, 1
    if skip is not None:
        a.skip = skip
    if out16 is not None:
        a.out16, a.out16_chunks_total, a.out16_chunk0 = out16, 6, out16_chunk0
    if out32 is not None:
        a.out32, a.out32_c = out32, 64
    L = _lib.lib()
    if bf16 or mask is not None:
        _lib.check(L.srbh_conv3x3_x16(C.byref(a), bf16, mask, 6, mask_chunk0, _lib.stream_ptr()), "conv3x3_x16")
    else:
        _lib.check(L.srbh_conv3x3_f16(C.byref(a), _lib.stream_ptr()), "conv3x3_f16")


TRUNK_BWD_PATHS = {"persistent": 0, "per_layer": 0}      # ... and the fast training backward
BWD_PERSISTENT = _os.environ.get("SRBH_SR_PTRUNK_BWD", "1") != "0"
TRUNK_WGRAD = _os.environ.get("SRBH_SR_TRUNK_WGRAD", "1") != "0"      # weight gradients of all RDBs as one launch (srbh_trunk_wgrad) behind the persistent backward
TRUNK_FWD_PATHS = {"persistent": 0, "per_layer": 0}      # which form the fast training forward took (tests / bench: no silent fallback)


def _trunk_fast_forward(net, feat):
    """feat (B,H,W,64) fp32 NHWC -> trunk output (same shape, fp32) + the saved dense buffers.  (The launch loop -- 5 convs per RDB,
    mirroring csrc/srbh_rrdbnet.hip's per-layer inference sequence -- runs behind ONE C-ABI call: at batch 8 the ~350 + ~2 000 launches of
    a step cost more in ctypes overhead than on the device.  `_conv16` above is the same call for tests / single layers.)"""
    L = _lib.lib()
    B, Hh, Ww, _ = feat.shape
    n_rdb = len(net.body) * 3
    ws = _fast_buffers(B, Hh, Ww, n_rdb, feat.device)
    _, desc = net._ensure_packed(feat.device)
    lease = _FastLease(ws)
    xr, xrr = feat.clone(), feat.clone()
    used = C.c_int(0)
    if ws.get("aux") is not None:
        # the 345 convs as ONE launch of the inference trunk's persistent kernel over the row of dense buffers (bit-identical to the sequence below)
        _lib.check(L.srbh_rrdbnet_trunk_train_forward_persistent(C.byref(desc), xr.data_ptr(), xrr.data_ptr(), ws["D"].data_ptr(), ws["nb"], B, Hh, Ww,
                                                                 ws["aux"].data_ptr(), _lib.stream_ptr(), C.byref(used)), "rrdbnet_trunk_train_forward_persistent")
    if not used.value:
        _lib.check(L.srbh_rrdbnet_trunk_train_forward(C.byref(desc), xr.data_ptr(), xrr.data_ptr(), ws["D"].data_ptr(), ws["nb"], B, Hh, Ww,
                                                      _lib.stream_ptr()), "rrdbnet_trunk_train_forward")
    TRUNK_FWD_PATHS["persistent" if used.value else "per_layer"] += 1
    return xr, lease


_DW_RDB = 9 * 26624          # weights of one RDB: 9 * (32*64 + 32*96 + 32*128 + 32*160 + 64*192)
_DW_OFF = (0, 9 * 2048, 9 * (2048 + 3072), 9 * (2048 + 3072 + 4096), 9 * (2048 + 3072 + 4096 + 5120))
_CONV_GEO = ((160, 32, 64), (128, 32, 96), (96, 32, 128), (64, 32, 160), (0, 64, 192))      # conv1..5: (G channel offset, cout, cin)


def _trunk_fast_backward(net, lease, g, grads):
    """g: gradient of the trunk output (B,H,W,64) fp32 contiguous -> gradient of its input; fills grads[id(param)]"""
    lease.check()
    ws = lease.ws
    L = _lib.lib()
    B, Hh, Ww, _ = g.shape
    packs = net.__dict__.setdefault("_srbh_trunk_bwd_packs", _TrunkBwdPacks()).get(net)
    n_rdb = len(net.body) * 3
    dev = g.device
    gb, gc = torch.empty_like(g), torch.empty_like(g)
    dw_all = torch.empty(n_rdb * _DW_RDB, dtype=torch.float32, device=dev)
    db_all = torch.empty(n_rdb * 192, dtype=torch.float32, device=dev)
    offs = (C.c_size_t * 5)(*packs.offs)
    gout = C.c_void_p()
    used = C.c_int(0)
    if BWD_PERSISTENT and ws.get("aux") is not None and ws["G"].numel() >= (n_rdb + 1) * ws["nb"]:
        # the 345 data-gradient convs as ONE launch of the persistent trunk kernel's bf16 form, the weight gradients of all RDBs behind it
        _lib.check(L.srbh_rrdbnet_trunk_train_backward_persistent(len(net.body), ws["D"].data_ptr(), ws["nb"], packs.buf.data_ptr(), packs.stride, offs,
                                                                  ws["zero_bias"].data_ptr(), g.data_ptr(), gb.data_ptr(), gc.data_ptr(), C.byref(gout),
                                                                  ws["G"].data_ptr(), ws["nb"], dw_all.data_ptr(), db_all.data_ptr(), ws["wg"].data_ptr(),
                                                                  ws["twg"].data_ptr() if (TRUNK_WGRAD and ws.get("twg") is not None) else None,
                                                                  B, Hh, Ww, ws["aux"].data_ptr(), _lib.stream_ptr(), C.byref(used)),
                   "rrdbnet_trunk_train_backward_persistent")
    TRUNK_BWD_PATHS["persistent" if used.value else "per_layer"] += 1
    if not used.value:
        _lib.check(L.srbh_rrdbnet_trunk_train_backward(len(net.body), ws["D"].data_ptr(), ws["nb"], packs.buf.data_ptr(), packs.stride, offs,
                                                   g.data_ptr(), gb.data_ptr(), gc.data_ptr(), C.byref(gout), ws["G"].data_ptr(), ws["nb"], dw_all.data_ptr(),
                                                   db_all.data_ptr(), ws["wg"].data_ptr(), B, Hh, Ww, _lib.stream_ptr()), "rrdbnet_trunk_train_backward")
    i = 0
    for blk in net.body:
        for r in (1, 2, 3):
            rdb = getattr(blk, f"rdb{r}")
            for k, (ch0, cout, cin) in enumerate(_CONV_GEO):
                conv = getattr(rdb, f"conv{k + 1}")
                o = i * _DW_RDB + _DW_OFF[k]
                grads[id(conv.weight)] = dw_all[o:o + cout * cin * 9].view(cout, cin, 3, 3)
                grads[id(conv.bias)] = db_all[i * 192 + ch0:i * 192 + ch0 + cout]
            i += 1
    return {g.data_ptr(): g, gb.data_ptr(): gb, gc.data_ptr(): gc}[gout.value]


class _Packs:
    """fp32 HWPACK32 images of one conv: forward pack (+ padded bias) and the transposed/flipped packs of its 64-channel
    input slices for the data gradient; rebuilt when the parameter changes."""

    def __init__(self):
        self.key = None

    def get(self, conv):
        w = conv.weight
        mixed = _mixed()
        key = (w._version, w.data_ptr(), conv.bias._version, conv.bias.data_ptr(), wcache.gen(w, conv.bias), mixed)
        if key != self.key:
            L = _lib.lib()
            cout, cin, ks, _ = w.shape
            wc = w.detach().float().contiguous()
            st = _lib.stream_ptr()
            self.mixed = mixed
            if mixed:          # fp16 forward pack, bf16 transposed / flipped packs for the data gradient (srbh_hpack_conv_h16)
                self.fwd = torch.empty(L.srbh_hpack_h16_bytes(cout, cin, ks) // 2, dtype=torch.float16, device=w.device)
                _lib.check(L.srbh_hpack_conv_h16(wc.data_ptr(), cout, cin, ks, 0, 0, self.fwd.data_ptr(), st), "hpack_h16(fwd)")
            else:
                self.fwd = torch.empty(L.srbh_hpack_bytes(cout, cin, ks) // 4, dtype=torch.float32, device=w.device)
                _lib.check(L.srbh_hpack_conv_f32(wc.data_ptr(), cout, cin, ks, 0, self.fwd.data_ptr(), st), "hpack(fwd)")
            self.bias = torch.zeros((cout + 15) // 16 * 16, dtype=torch.float32, device=w.device)
            self.bias[:cout] = conv.bias.detach().float()
            self.bwd = []          # [(c_lo, n, pack)]: data-gradient conv producing input channels c_lo .. c_lo + n
            for c_lo in range(0, cin, 64):
                n = min(64, cin - c_lo)
                sub = wc[:, c_lo:c_lo + n].contiguous()
                if mixed:
                    pk = torch.empty(L.srbh_hpack_h16_bytes(n, cout, ks) // 2, dtype=torch.float16, device=w.device)
                    _lib.check(L.srbh_hpack_conv_h16(sub.data_ptr(), n, cout, ks, 1, 1, pk.data_ptr(), st), "hpack_h16(bwd)")
                else:
                    pk = torch.empty(L.srbh_hpack_bytes(n, cout, ks) // 4, dtype=torch.float32, device=w.device)
                    _lib.check(L.srbh_hpack_conv_f32(sub.data_ptr(), n, cout, ks, 1, pk.data_ptr(), st), "hpack(bwd)")
                self.bwd.append((c_lo, n, pk))
            self.key = key
        wcache.keep(self.fwd, self.bias, self.bwd)
        return self


def _packs(conv) -> _Packs:
    return conv.__dict__.setdefault("_srbh_train_packs", _Packs()).get(conv)


def _conv(src, c0, ld0, pack, bias, cout, out, out_ld=0, out_coff=0, lrelu=False, res1=None, res2=None, bf16=False):
    """out[..., out_coff : out_coff + cout] = epilogue(conv3x3(src[..., :c0])): the exact-fp32 kernel, or -- `pack` is a 16-bit pack
    (mixed precision) -- the fp16 (forward) / bf16 (`bf16`: data gradients) operand form with the same fp32 epilogue."""
    a = _lib.HConvArgs()
    a.src0, a.c0, a.src0_ld = src.data_ptr(), c0, ld0
    a.w, a.bias = pack.data_ptr(), (bias.data_ptr() if bias is not None else None)
    a.cout, a.ksize = cout, 3
    a.B, a.H, a.W = src.shape[0], src.shape[1], src.shape[2]
    a.out, a.out_ld, a.out_coff, a.post_lrelu = out.data_ptr(), out_ld, out_coff, int(lrelu)
    if res1 is not None:
        a.res1, a.res1_ld, a.res1_scale = res1[0].data_ptr(), res1[1], res1[2]
    if res2 is not None:
        a.res2, a.res2_ld, a.res2_scale = res2[0].data_ptr(), res2[1], res2[2]
    if pack.dtype == torch.float16:
        _lib.check(_lib.lib().srbh_hconv_h16(C.byref(a), int(bf16), _lib.stream_ptr()), "hconv_h16(rrdbnet train)")
    else:
        _lib.check(_lib.lib().srbh_hconv_f32(C.byref(a), _lib.stream_ptr()), "hconv_f32(rrdbnet train)")


def _wgrad(src, c0, ld0, g, cout):
    """dW[cout][c0][3][3] = sum_px g[px][oc] * src[px + tap][ci]; src may be a strided view (ld0 floats per pixel)."""
    L = _lib.lib()
    dw = torch.empty((cout, c0, 3, 3), dtype=torch.float32, device=g.device)
    a = _lib.HWGradArgs()
    a.src0, a.c0, a.src0_ld = src.data_ptr(), c0, ld0
    a.dy, a.cout, a.ksize = g.data_ptr(), cout, 3
    a.B, a.H, a.W = g.shape[0], g.shape[1], g.shape[2]
    a.dw = dw.data_ptr()
    ws = torch.empty(L.srbh_hwgrad_ws_bytes(cout, c0, 3) // 4, dtype=torch.float32, device=g.device)
    a.ws = ws.data_ptr()
    if _mixed():           # (layers outside the bf16 kernel's granularity -- conv_first's 3 input channels -- are computed in fp32 inside)
        _lib.check(L.srbh_hconv_wgrad_b16(C.byref(a), _lib.stream_ptr()), "hconv_wgrad_b16(rrdbnet train)")
    else:
        _lib.check(L.srbh_hconv_wgrad_f32(C.byref(a), _lib.stream_ptr()), "hconv_wgrad_f32(rrdbnet train)")
    return dw


def _dgrad_into(g, cout, packs, dst, dst_ld, accumulate):
    """dst[..., :cin] (+)= conv^T(g, W), 64 input channels per launch; `accumulate`: add to what dst holds (residual epilogue)."""
    for c_lo, n, pk in packs.bwd:
        _conv(g, cout, cout, pk, None, n, dst, dst_ld, c_lo, res1=(dst[..., c_lo:], dst_ld, 1.0) if accumulate else None, bf16=True)


def _up2(t):
    Bn, Hn, Wn, Cn = t.shape
    o = torch.empty((Bn, 2 * Hn, 2 * Wn, Cn), dtype=torch.float32, device=t.device)
    _lib.check(_lib.lib().srbh_nearest2x_f32(t.data_ptr(), o.data_ptr(), Bn, 2 * Hn, 2 * Wn, Cn, _lib.stream_ptr()), "nearest2x")
    return o


def _up2_bwd(g):            # adjoint of nearest x2: the sum over each 2x2 block
    Bn, Hn, Wn, Cn = g.shape
    if g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and Cn % 4 == 0:
        o = torch.empty((Bn, Hn // 2, Wn // 2, Cn), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().srbh_up2_bwd_nhwc_f32(g.data_ptr(), o.data_ptr(), Bn, Hn // 2, Wn // 2, Cn, _lib.stream_ptr()), "up2_bwd_nhwc_f32")
        return o
    return g.reshape(Bn, Hn // 2, 2, Wn // 2, 2, Cn).sum(dim=(2, 4))


def _bias_grad(g):          # (B,H,W,C) -> (C,): the head's plane-sum kernels (fp64 partial sums) for the 64-channel gradients
    if g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape[3] % 16 == 0:
        from . import hrfuse_autograd as HA
        return HA.channel_sum(g.permute(0, 3, 1, 2))
    return g.sum(dim=(0, 1, 2))


def _lrelu_mask_(g, y):     # g *= d lrelu(z)/dz with y = lrelu(z): y > 0 <=> z > 0 (torch: slope at z <= 0)
    if (g.is_cuda and g.dtype == torch.float32 and y.dtype == torch.float32 and g.is_contiguous() and y.is_contiguous() and g.shape == y.shape
            and g.numel() % 4 == 0):
        # one pass (srbh_lrelu_bwd_f32) instead of three stock element-wise kernels over 256 x 256 x 64 fp32 tensors
        _lib.check(_lib.lib().srbh_lrelu_bwd_f32(g.data_ptr(), y.data_ptr(), 0.2, g.numel(), _lib.stream_ptr()), "lrelu_bwd_f32")
        return g
    return g.mul_(torch.where(y > 0, 1.0, 0.2))


class _RRDBNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, want_forward, *params):
        dev = x.device
        xs = x.detach().float().permute(0, 2, 3, 1).contiguous()          # (B,H,W,Cin) NHWC
        B, Hh, Ww, cin = xs.shape
        convs = {n: _packs(getattr(net, n)) for n in ("conv_first", "conv_body", "conv_up1", "conv_up2", "conv_hr", "conv_last")}
        feat = torch.empty((B, Hh, Ww, 64), dtype=torch.float32, device=dev)
        _conv(xs, cin, cin, convs["conv_first"].fwd, convs["conv_first"].bias, 64, feat)
        dense = []                                                          # one saved (B,H,W,192) buffer per RDB
        cur = feat
        fast_ws = None
        if _fast() and Hh % 8 == 0 and Ww % 64 == 0 and len(net.body) > 0:
            cur, fast_ws = _trunk_fast_forward(net, feat)
        for blk in (net.body if fast_ws is None else ()):
            x_rrdb = cur
            for r in (1, 2, 3):
                rdb = getattr(blk, f"rdb{r}")
                D = torch.empty((B, Hh, Ww, 192), dtype=torch.float32, device=dev)
                D[..., :64].copy_(cur)
                for k in range(1, 5):
                    p = _packs(getattr(rdb, f"conv{k}"))
                    c0 = 64 + 32 * (k - 1)
                    _conv(D, c0, 192, p.fwd, p.bias, 32, D, 192, c0, lrelu=True)
                p = _packs(rdb.conv5)
                nxt = torch.empty((B, Hh, Ww, 64), dtype=torch.float32, device=dev)
                _conv(D, 192, 192, p.fwd, p.bias, 64, nxt, res1=(D, 192, 0.2), res2=(x_rrdb, 64, 0.2) if r == 3 else None)
                dense.append(D)
                cur = nxt
        body = torch.empty_like(feat)
        _conv(cur, 64, 64, convs["conv_body"].fwd, convs["conv_body"].bias, 64, body, res1=(feat, 64, 1.0))
        t1 = _up2(body)
        u1 = torch.empty_like(t1)
        _conv(t1, 64, 64, convs["conv_up1"].fwd, convs["conv_up1"].bias, 64, u1, lrelu=True)
        t2 = _up2(u1)
        u2 = torch.empty_like(t2)
        _conv(t2, 64, 64, convs["conv_up2"].fwd, convs["conv_up2"].bias, 64, u2, lrelu=True)
        hr = torch.empty_like(u2)
        _conv(u2, 64, 64, convs["conv_hr"].fwd, convs["conv_hr"].bias, 64, hr, lrelu=bool(want_forward))
        ctx.net, ctx.want_forward, ctx.in_shape = net, bool(want_forward), tuple(x.shape)
        ctx.saved = (xs, feat, dense, cur, t1, u1, t2, u2, hr)
        ctx.fast_ws = fast_ws
        if not want_forward:
            return hr.permute(0, 3, 1, 2)
        cout = net.conv_last.out_channels
        out = torch.empty((B, 4 * Hh, 4 * Ww, cout), dtype=torch.float32, device=dev)
        _conv(hr, 64, 64, convs["conv_last"].fwd, convs["conv_last"].bias, cout, out)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        xs, feat, dense, last, t1, u1, t2, u2, hr = ctx.saved
        grads = {}

        def conv_bwd(name_or_mod, src, c0, ld0, g, need_dx=True, dst=None, dst_ld=0, accumulate=False):
            """records dW / db of the conv; returns (or accumulates into dst) the gradient of its input"""
            mod = getattr(net, name_or_mod) if isinstance(name_or_mod, str) else name_or_mod
            p = _packs(mod)
            cout = mod.out_channels
            grads[id(mod.weight)] = _wgrad(src, c0, ld0, g, cout)
            grads[id(mod.bias)] = _bias_grad(g)
            if not need_dx:
                return None
            if dst is None:
                dst = torch.empty(src.shape[:3] + (c0,), dtype=torch.float32, device=g.device)
                dst_ld = c0
            _dgrad_into(g, cout, p, dst, dst_ld, accumulate)
            return dst

        g = gout.detach().float().permute(0, 2, 3, 1).contiguous()
        if ctx.want_forward:
            g = conv_bwd("conv_last", hr, 64, 64, g)
            _lrelu_mask_(g, hr)
        g = conv_bwd("conv_hr", u2, 64, 64, g)
        _lrelu_mask_(g, u2)
        g = _up2_bwd(conv_bwd("conv_up2", t2, 64, 64, g))
        _lrelu_mask_(g, u1)
        g_feat2 = _up2_bwd(conv_bwd("conv_up1", t1, 64, 64, g))           # gradient of feat + body_feat
        g = conv_bwd("conv_body", last, 64, 64, g_feat2)                  # ... flows into the body's output
        i = len(dense)
        fast_ws = getattr(ctx, "fast_ws", None)
        if fast_ws is not None:
            g = _trunk_fast_backward(net, fast_ws, g.contiguous(), grads)
            if not getattr(ctx, "_srbh_retain", False):
                # the saved planes are free for the next forward as soon as the launches above are enqueued (same stream: the next
                # forward's writes are ordered behind this backward's reads); backward(retain_graph=True) callers keep the lease
                # by setting ctx._srbh_retain (not used by the trainer)
                fast_ws.release()
        for blk in (reversed(list(net.body)) if fast_ws is None else ()):
            g_rrdb_out = g                                                # out = rdb3(.)*0.2 + x_rrdb
            g = g_rrdb_out * 0.2
            for r in (3, 2, 1):
                rdb = getattr(blk, f"rdb{r}")
                i -= 1
                D = dense[i]
                # x_next = conv5(D) * 0.2 + x : the conv sees 0.2 g, x (= D[..., :64]) sees g
                dD = torch.zeros_like(D)
                g5 = (g * 0.2).contiguous()
                conv_bwd(rdb.conv5, D, 192, 192, g5, dst=dD, dst_ld=192, accumulate=False)
                for k in (4, 3, 2, 1):
                    c0 = 64 + 32 * (k - 1)
                    gk = _lrelu_mask_(dD[..., c0:c0 + 32].contiguous(), D[..., c0:c0 + 32])
                    conv_bwd(getattr(rdb, f"conv{k}"), D, c0, 192, gk, dst=dD, dst_ld=192, accumulate=True)
                g = dD[..., :64] + g
            g = g + g_rrdb_out                                            # the RRDB's skip connection
        g = g + g_feat2                                                   # feat reaches the output through the trunk skip too
        need_dx = ctx.needs_input_grad[0]
        gx = conv_bwd("conv_first", xs, xs.shape[3], xs.shape[3], g.contiguous(), need_dx=need_dx)
        params = [p for p in net.parameters()]
        out = [gx.permute(0, 3, 1, 2).reshape(ctx.in_shape) if need_dx else None, None, None]
        for p in params:
            gp = grads.get(id(p))
            out.append(None if gp is None else gp.reshape(p.shape).to(p.dtype))
        return tuple(out)


def rrdbnet_apply(net, x, want_forward):
    """RRDBNet.forward / forward_feature with a recorded graph (x: (B,C,H,W) device tensor, any scale handled by the caller)."""
    params = list(net.parameters())
    return _RRDBNetFn.apply(x, net, bool(want_forward), *params)
