"""GPU parity of the fused 3x3 convolution kernel (libsrbh srbh_conv3x3_f16) through the C ABI.

Oracle = CPU conv of the SAME fp16-rounded operands accumulated in float64: the kernel accumulates in
fp32 on the matrix cores, so agreement must be ~1e-6 (summation order only); against unrounded fp32
operands the error is the fp16 operand rounding, bounded here at 2e-3 per layer."""
import pytest
import torch

from oracle import srbh_oracle as O
from tests import gpu_util as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TIGHT = 5e-6


def rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


@pytest.mark.parametrize("B,cin,cout,H,W", [
    (2, 64, 32, 64, 64),      # RDB conv1
    (1, 96, 32, 64, 64),      # conv2
    (1, 160, 32, 16, 64),     # conv4, short image
    (2, 192, 64, 64, 64),     # conv5 shape (plain epilogue)
    (1, 64, 64, 24, 40),      # ragged: H,W not multiples of the 8x64 tile
    (3, 32, 32, 7, 5),        # tiny, one chunk
    (1, 64, 64, 8, 200),      # several tiles along x, ragged last one
])
def test_conv_plain_lrelu_out16(B, cin, cout, H, W):
    x, w, b = rnd((B, cin, H, W), 1), rnd((cout, cin, 3, 3), 2, -0.1, 0.1), rnd((cout,), 3)
    xin = G.act16_from_nchw(x.to(DEV))
    out = G.act16_alloc(B, cout // 32, H, W, DEV)
    wp, bias_dev = G.pack_w(w.to(DEV)), b.to(DEV)
    a = G.conv_args(**{"in": xin.data_ptr()}, in_chunks_total=cin // 32, in_chunk0=0, in_chunks=cin // 32,
                    w=wp.data_ptr(), bias=bias_dev.data_ptr(), cout=cout, B=B, H=H, W=W, lrelu=1,
                    out16=out.data_ptr(), out16_chunks_total=cout // 32, out16_chunk0=0)
    G.run_conv(a)
    got = G.act16_to_nchw(out, B, cout, H, W).cpu()
    want = torch.nn.functional.leaky_relu(G.ref_conv(x, w, b), 0.2)
    assert O.rel_l2(got, G.h16(want)) <= 3e-4          # output itself is stored as fp16
    assert O.max_rel(got, G.h16(want)) <= 1e-3
    full = torch.nn.functional.leaky_relu(G.ref_conv(x, w, b, rounded=False), 0.2)
    assert O.rel_l2(got, full) <= 2e-3
    # the zero border of the output buffer must be untouched
    raw = out.view(torch.float16)[: B * (cout // 32) * (H + 2) * (W + 2) * 32].view(B, cout // 32, H + 2, W + 2, 32)
    assert float(raw[:, :, 0].abs().max()) == 0 and float(raw[:, :, -1].abs().max()) == 0
    assert float(raw[:, :, :, 0].abs().max()) == 0 and float(raw[:, :, :, -1].abs().max()) == 0


@pytest.mark.parametrize("B,H,W,oc", [(2, 16, 64, 64), (1, 24, 72, 64), (1, 8, 8, 3)])
def test_conv_out32_nhwc(B, H, W, oc):
    cin, cout = 64, 64 if oc == 64 else 32
    x, w, b = rnd((B, cin, H, W), 4), rnd((oc, cin, 3, 3), 5, -0.1, 0.1), rnd((oc,), 6)
    xin = G.act16_from_nchw(x.to(DEV))
    out = torch.full((B, H, W, oc), 7.0, device=DEV)
    wp = G.pack_w(w.to(DEV))
    bp = torch.zeros(cout, device=DEV)
    bp[:oc] = b.to(DEV)
    a = G.conv_args(**{"in": xin.data_ptr()}, in_chunks_total=2, in_chunk0=0, in_chunks=2, w=wp.data_ptr(),
                    bias=bp.data_ptr(), cout=cout, B=B, H=H, W=W, out32=out.data_ptr(), out32_c=oc)
    G.run_conv(a)
    got = out.permute(0, 3, 1, 2).cpu()
    assert O.rel_l2(got, G.ref_conv(x, w, b)) <= TIGHT
    assert O.rel_l2(got, G.ref_conv(x, w, b, rounded=False)) <= 2e-3


@pytest.mark.parametrize("B,H,W", [(1, 16, 128), (2, 8, 64), (1, 20, 36)])
def test_conv_nearest2x_fold(B, H, W):
    """conv(F.interpolate(x, 2, 'nearest')) with the index map folded into the LDS addressing."""
    cin = cout = 64
    x, w, b = rnd((B, cin, H // 2, W // 2), 7), rnd((cout, cin, 3, 3), 8, -0.1, 0.1), rnd((cout,), 9)
    xin = G.act16_from_nchw(x.to(DEV))
    out = torch.zeros((B, H, W, cout), device=DEV)
    wp, bp = G.pack_w(w.to(DEV)), b.to(DEV)
    a = G.conv_args(**{"in": xin.data_ptr()}, in_chunks_total=2, in_chunk0=0, in_chunks=2, w=wp.data_ptr(),
                    bias=bp.data_ptr(), cout=cout, B=B, H=H, W=W, upsample2x=1, out32=out.data_ptr(), out32_c=64)
    G.run_conv(a)
    got = out.permute(0, 3, 1, 2).cpu()
    assert O.rel_l2(got, G.ref_conv(x, w, b, ups=True)) <= TIGHT


def test_nearest2x_index_map_bit_exact():
    """identity centre-tap filter: the kernel output must equal the oracle's gather exactly (fp16-exact ints)."""
    B, H, W = 1, 16, 64
    x = torch.randint(-512, 512, (B, 64, H // 2, W // 2)).float()
    w = torch.zeros(64, 64, 3, 3)
    w[torch.arange(64), torch.arange(64), 1, 1] = 1.0
    xin = G.act16_from_nchw(x.to(DEV))
    out = torch.zeros((B, H, W, 64), device=DEV)
    wp = G.pack_w(w.to(DEV))
    a = G.conv_args(**{"in": xin.data_ptr()}, in_chunks_total=2, in_chunk0=0, in_chunks=2, w=wp.data_ptr(), bias=None,
                    cout=64, B=B, H=H, W=W, upsample2x=1, out32=out.data_ptr(), out32_c=64)
    G.run_conv(a)
    assert torch.equal(out.permute(0, 3, 1, 2).cpu(), O.nearest2x(x))


def test_conv_dense_chunk_offsets_and_residual_epilogues():
    """conv5-style call: reads 6 planes, x5*0.2+x into res1 (in place), then *0.2 + res2, fp16 copy to another buffer."""
    B, H, W = 2, 16, 64
    x = rnd((B, 192, H, W), 10)
    w, b = rnd((64, 192, 3, 3), 11, -0.05, 0.05), rnd((64,), 12)
    r1, r2 = rnd((B, 64, H, W), 13), rnd((B, 64, H, W), 14)
    xin = G.act16_from_nchw(x.to(DEV))
    nxt = G.act16_alloc(B, 6, H, W, DEV)
    res1 = r1.permute(0, 2, 3, 1).contiguous().to(DEV)
    res2 = r2.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp, bp = G.pack_w(w.to(DEV)), b.to(DEV)
    a = G.conv_args(**{"in": xin.data_ptr()}, in_chunks_total=6, in_chunk0=0, in_chunks=6, w=wp.data_ptr(),
                    bias=bp.data_ptr(), cout=64, B=B, H=H, W=W, res_scale=0.2, res1=res1.data_ptr(), res1_update=1,
                    res2_scale=0.2, res2=res2.data_ptr(), res2_update=1, out16=nxt.data_ptr(), out16_chunks_total=6,
                    out16_chunk0=0)
    G.run_conv(a)
    want = (G.ref_conv(x, w, b) * 0.2 + r1) * 0.2 + r2
    assert O.rel_l2(res1.permute(0, 3, 1, 2).cpu(), want) <= TIGHT
    assert O.rel_l2(res2.permute(0, 3, 1, 2).cpu(), want) <= TIGHT
    assert O.rel_l2(G.act16_to_nchw(nxt, B, 192, H, W)[:, :64].cpu(), G.h16(want)) <= 3e-4
    # rdb1/rdb2 flavour: only res1, and a middle-plane output (conv3 writes plane 4 of the same buffer it reads)
    res1b = r1.permute(0, 2, 3, 1).contiguous().to(DEV)
    a2 = G.conv_args(**{"in": xin.data_ptr()}, in_chunks_total=6, in_chunk0=0, in_chunks=6, w=wp.data_ptr(),
                     bias=bp.data_ptr(), cout=64, B=B, H=H, W=W, res_scale=0.2, res1=res1b.data_ptr(), res1_update=1,
                     out16=nxt.data_ptr(), out16_chunks_total=6, out16_chunk0=2)
    G.run_conv(a2)
    want2 = G.ref_conv(x, w, b) * 0.2 + r1
    assert O.rel_l2(res1b.permute(0, 3, 1, 2).cpu(), want2) <= TIGHT
    planes = G.act16_to_nchw(nxt, B, 192, H, W).cpu()
    assert O.rel_l2(planes[:, 64:128], G.h16(want2)) <= 3e-4
    assert O.rel_l2(planes[:, :64], G.h16(want)) <= 3e-4   # earlier planes untouched
    # skip flavour (conv_body): y = conv + feat
    wb = rnd((64, 64, 3, 3), 15, -0.1, 0.1)
    wpb = G.pack_w(wb.to(DEV))
    feat = rnd((B, 64, H, W), 16)
    featd = feat.permute(0, 2, 3, 1).contiguous().to(DEV)
    o32 = torch.zeros((B, H, W, 64), device=DEV)
    a3 = G.conv_args(**{"in": xin.data_ptr()}, in_chunks_total=6, in_chunk0=2, in_chunks=2, w=wpb.data_ptr(),
                     bias=bp.data_ptr(), cout=64, B=B, H=H, W=W, skip=featd.data_ptr(), out32=o32.data_ptr(), out32_c=64)
    G.run_conv(a3)
    assert O.rel_l2(o32.permute(0, 3, 1, 2).cpu(), G.ref_conv(x[:, 64:128], wb, b) + feat) <= TIGHT


def test_argument_errors_are_loud():
    from srbh_amd import _lib
    a = G.conv_args(cout=48, B=1, H=8, W=8)
    with pytest.raises(RuntimeError, match="null|cout"):
        G.run_conv(a)
    x = G.act16_alloc(1, 2, 8, 8, DEV)
    a = G.conv_args(**{"in": x.data_ptr()}, in_chunks_total=2, in_chunk0=1, in_chunks=2, w=x.data_ptr(), cout=32, B=1,
                    H=8, W=8, out16=x.data_ptr(), out16_chunks_total=2)
    with pytest.raises(RuntimeError, match="chunk range"):
        G.run_conv(a)
    assert b"chunk" in _lib.lib().srbh_last_error()


@pytest.mark.parametrize("ups,H,W", [(1, 256, 256), (0, 256, 256), (0, 250, 200)])
def test_persistent_tail_form_is_bit_identical(ups, H, W, monkeypatch):
    """>= 512 tiles of a 64 -> 64 conv take the persistent form (srbh_ptail.hip): same bits as the per-tile kernel, for
    the nearest-x2 + lrelu + fp16 output (conv_up*) and the plain fp32 NHWC output (conv_hr), incl. ragged edges."""
    B, cin, cout = 5, 64, 64
    hin, win = (H // 2, W // 2) if ups else (H, W)
    x, w, b = rnd((B, cin, hin, win), 21), rnd((cout, cin, 3, 3), 22, -0.1, 0.1), rnd((cout,), 23)
    xin = G.act16_from_nchw(x.to(DEV))
    wp, bp = G.pack_w(w.to(DEV)), b.to(DEV)
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("SRBH_PTAIL", mode)
        o16 = G.act16_alloc(B, 2, H, W, DEV)
        o32 = torch.zeros((B, H, W, cout), device=DEV)
        a = G.conv_args(**{"in": xin.data_ptr()}, in_chunks_total=2, in_chunk0=0, in_chunks=2, w=wp.data_ptr(),
                        bias=bp.data_ptr(), cout=cout, B=B, H=H, W=W, upsample2x=ups, lrelu=ups,
                        out16=o16.data_ptr(), out16_chunks_total=2, out16_chunk0=0, out32=o32.data_ptr(), out32_c=64)
        G.run_conv(a)
        torch.cuda.synchronize()
        res.append((o16.clone(), o32.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    want = G.ref_conv(x[:1], w, b, ups=bool(ups))
    if ups:
        want = torch.nn.functional.leaky_relu(want, 0.2)
    assert O.rel_l2(res[0][1][:1].permute(0, 3, 1, 2).cpu(), want) <= TIGHT
