"""developer timing: srbh_hbwd16 (one pass) against the three launches it replaces, at the training step's shape (B x 16 x 256 x 256).
python tools/time_hbwd16.py [B]     (SRBH_HBWD16_WGS=... to vary the grid)"""
import sys
import torch
sys.path.insert(0, '.')
from srbh_amd import _lib
from srbh_amd import hrfuse as H
from srbh_amd import hrfuse_autograd as HA
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(1)
nh = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)      # noqa: E731
gy = nh(torch.randn((B, 16, 256, 256), generator=g) * 1e-3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
c, x = nh(torch.randn((B, 16, 256, 256), generator=g)), nh(torch.randn((B, 16, 256, 256), generator=g))
res = gy.clone()
v = lambda s=1.0: (torch.rand(16, generator=g) * s + 0.5).to(dev)      # noqa: E731
mean, invstd, consts = v(), v(), (v(), v(1e-5), v(1e-5))
s1, h1, m1, i1 = v(), v(), v(), v()
w = (torch.randn((16, 16, 3, 3), generator=g) * 0.1).to(dev)
pg = HA._PackedGrad()
L = _lib.lib()


def T(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def sep(mask, pre, res_, out_b16, bstat):
    dc = H.empty_nhwc(B, 16, 256, 256, dev, torch.bfloat16)
    ms, mh = (mask[0].data_ptr(), mask[1].data_ptr()) if mask is not None else (None, None)
    _lib.check(L.srbh_bn_bwd_apply_io(gy.data_ptr(), c.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ms, mh, consts[0].data_ptr(), consts[1].data_ptr(),
                                      consts[2].data_ptr(), dc.data_ptr(), B * 65536, 16, 5, _lib.stream_ptr()))
    HA.conv_wgrad([x], pre, dc, 16, 3)
    HA.conv_dgrad(dc, w, pg, res=res_, out_b16=out_b16, bstat=bstat)


px = B * 65536
with H.head_precision("f16"), torch.no_grad():
    st = HA._stats_buf(16, dev)
    t1 = T(lambda: HA.hbwd16(gy, c, mean, invstd, consts, None, x, (s1, h1, True), w, pg, out_b16=True, bstat=(x, m1, i1, s1, h1, st)))
    t1s = T(lambda: sep(None, (s1, h1, True), None, True, (x, m1, i1, s1, h1, st)))
    t2 = T(lambda: HA.hbwd16(gy, c, mean, invstd, consts, (s1, h1), x, None, w, pg, res=res, out_b16=False))
    t2s = T(lambda: sep((s1, h1), None, res, False, None))
    # the chain hand-off form: conv1's backward also does the previous block's reduce pass (masked bf16 out + bn2' sums)
    out0, bits = H.bn_add_relu(c, s1, h1, x, want_bits=True)
    st2 = HA._stats_buf(16, dev)
    t3 = T(lambda: HA.hbwd16(gy, c, mean, invstd, consts, (s1, h1), x, None, w, pg, res=res, out_b16=True, bstat=(c, m1, i1, None, None, st2), relu_bits=bits))
    gfp = torch.randn_like(c)

    def reduce_pass():
        HA.bn_backward(gfp, c, m1, i1, v(), None, True, relu_ref=bits, out_b16=True, apply=False)
    t3r = T(reduce_pass)
print(f"B={B} conv1 form + the previous block's reduce pass: fused {t3:.1f} us | conv1 form {t2:.1f} + reduce pass {t3r:.1f} = {t2 + t3r:.1f} us")
print(f"B={B} conv2 form: fused {t1:.1f} us ({px * 256 / t1 / 1e3:.0f} GB/s on 256 B/px incl. the epilogue's c) | three launches {t1s:.1f} us")
print(f"B={B} conv1 form: fused {t2:.1f} us ({px * 256 / t2 / 1e3:.0f} GB/s on 256 B/px) | three launches {t2s:.1f} us")
