"""CPU: the drop-in boundary -- state_dict layout, constructor signatures, C-ABI symbol table,
loud failure without a device.  No kernel is launched here."""
import inspect
import os
import re

import pytest
import torch

from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from srbh_amd import _lib
    _lib.build()
    lib = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "srbh.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(srbh_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in srbh.h but not exported by libsrbh.so"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.srbh_version() >= 100
    # pure host-side size queries
    assert lib.srbh_wpack16_bytes(32, 64) == 2 * 18 * 1024
    assert lib.srbh_wpack16_bytes(64, 192) == 6 * 18 * 2 * 1024
    assert lib.srbh_wpack16_bytes(3, 64) == 2 * 18 * 1024  # cout padded to 32
    assert lib.srbh_act16_bytes(1, 64, 64, 64) >= 2 * 66 * 66 * 64
    assert lib.srbh_rrdbnet_workspace_bytes(1, 64, 64, 0) > 0
    assert lib.srbh_rrdbnet_workspace_bytes(1, 64, 64, 1) > lib.srbh_rrdbnet_workspace_bytes(1, 64, 64, 0)
    assert lib.srbh_rrdbnet_workspace_bytes(0, 64, 64, 0) == 0


def _header_fields(struct):
    hdr = open(os.path.join(ROOT, "include", "srbh.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"([A-Za-z_0-9]+)\s*$", part.strip())[0])
    return names


def test_ctypes_struct_matches_header_field_order():
    """every argument struct of include/srbh.h against its ctypes mirror in _lib.py, field for field"""
    from srbh_amd import _lib
    pairs = {"srbh_conv3x3_args": _lib.ConvArgs, "srbh_hconv_args": _lib.HConvArgs, "srbh_hwgrad_args": _lib.HWGradArgs,
             "srbh_bnact_args": _lib.BnActArgs, "srbh_bnact_bwd_args": _lib.BnActBwdArgs, "srbh_conv_w": _lib.ConvW,
             "srbh_rrdbnet_desc": _lib.RRDBNetDesc, "srbh_mbmid_args": _lib.MbMidArgs, "srbh_mbmid_bwd_args": _lib.MbMidBwdArgs,
             "srbh_hbwd16_args": _lib.HBwd16Args}
    for struct, cls in pairs.items():
        assert _header_fields(struct) == [n.rstrip("_") for n, _ in cls._fields_], struct
    assert _header_fields("srbh_transpose_desc") == ["src", "dst", "rows", "cols"]       # (built as a numpy record in encoders.py)
    assert _header_fields("srbh_dconv_pack_desc") == ["w", "fwd", "bwd", "cout", "cin"]   # (ditto: encoders.DecoderPackTable)
    from srbh_amd import optim
    assert _header_fields("srbh_adam_entry") == list(optim._ENTRY.names) and optim._ENTRY.itemsize == 56        # (numpy record in optim.Adam)


def test_rrdbnet_state_dict_layout_and_signature():
    from srbh_amd.rrdbnet import RRDBNet, RRDB, ResidualDenseBlock, make_layer
    net = RRDBNet(3, 3)
    sd = net.state_dict()
    want = synth.rrdbnet_state_dict()  # validated strict=True against the reference in tools/make_golden.py
    assert list(sd.keys()) == list(want.keys()) or set(sd.keys()) == set(want.keys())
    for k, v in want.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert len(sd) == 702 and sum(p.numel() for p in net.parameters()) == 16_697_987
    assert net.scale == 4
    net.load_state_dict(want, strict=True)
    sig = inspect.signature(RRDBNet.__init__)
    assert list(sig.parameters)[1:] == ["num_in_ch", "num_out_ch", "scale", "num_feat", "num_block", "num_grow_ch"]
    assert [sig.parameters[k].default for k in ("scale", "num_feat", "num_block", "num_grow_ch")] == [4, 64, 23, 32]
    assert RRDBNet(3, 3, scale=2, num_block=1).conv_first.in_channels == 12
    assert RRDBNet(3, 3, scale=1, num_block=1).conv_first.in_channels == 48
    assert isinstance(make_layer(RRDB, 2, num_feat=64, num_grow_ch=32), torch.nn.Sequential)
    # RDB convs follow default_init_weights(.., 0.1): zero bias, std = 0.1*sqrt(2/fan_in)
    rdb = ResidualDenseBlock(64, 32)
    assert float(rdb.conv3.bias.abs().max()) == 0.0
    std = float(rdb.conv5.weight.std())
    assert abs(std - 0.1 * (2.0 / (192 * 9)) ** 0.5) / std < 0.05


def test_no_cpu_fallback():
    from srbh_amd.rrdbnet import RRDBNet, ResidualDenseBlock, pixel_unshuffle
    net = RRDBNet(3, 3, num_block=1)
    with pytest.raises(RuntimeError, match="no CPU"):
        net.forward_feature(torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError):
        ResidualDenseBlock()(torch.zeros(1, 64, 8, 8))
    with pytest.raises(AssertionError):
        pixel_unshuffle(torch.zeros(1, 1, 5, 4), 2)
    x = torch.arange(1 * 3 * 8 * 12, dtype=torch.float32).reshape(1, 3, 8, 12)
    from oracle import srbh_oracle as O
    assert torch.equal(pixel_unshuffle(x, 2), O.pixel_unshuffle(x, 2))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "super-resolution-building-height-estimation_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), os.path.join(dp, fn)
                assert "/root/reference" not in txt or fn.endswith(".md"), os.path.join(dp, fn)
