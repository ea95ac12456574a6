#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of the head kernels (tools/head_kernels.py under rocprofv3 --pmc, one pass per counter) ->
profiles/<tag>_pmc_head_kernels.json.  usage: pmc_head.py <fetch_dir> <write_dir> <out.json> [B]
Units as tools/pmc_traffic.py: KiB as reported, FETCH_SIZE doubled (MI355X_MICROARCH.md #HBM, gfx950 correction)."""
import json, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import load

fetch_dir, write_dir, out = sys.argv[1:4]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
px = B * 256 * 256
alg = {"hconv16_kernel<1>": ("forward 16->16 3x3, fp16 operands, + BN statistics (persistent kernel)", px * 128), "hconv16_kernel<2>": ("data gradient 16->16 3x3, bf16 operands (persistent kernel)", px * 128),
       "hconv_f32_kernel<1, 3, 1, 1>": ("forward 16->16 3x3, fp16 operands, + BN statistics", px * 128), "hconv_f32_kernel<1, 3, 1, 2>": ("data gradient 16->16 3x3, bf16 operands", px * 128),
       "hwgrad16_kernel": ("weight gradient 16->16 3x3, bf16 operands (double-buffered kernel)", px * 128),
       "hwgrad_b16_kernel<3>": ("weight gradient 16->16 3x3, bf16 operands", px * 128), "hwgrad_f32_kernel<3>": ("weight gradient 16->16 3x3, fp32", px * 128),
       "bn_bwd_reduce4_kernel": ("BatchNorm backward: per-channel sums of dy, dy*xhat (16-byte form)", px * 128), "bn_bwd_apply4_kernel": ("BatchNorm backward: dx (16-byte form)", px * 192),
       "bn_bwd_reduce_kernel": ("BatchNorm backward: per-channel sums of dy, dy*xhat", px * 128), "bn_bwd_apply_kernel": ("BatchNorm backward: dx", px * 192),
       "bn_add_relu_kernel": ("relu(bn(c) + identity)", px * 192)}
res = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python tools/head_kernels.py %d (separate passes)" % B,
       "units": "bytes per launch; FETCH_SIZE doubled (MI355X_MICROARCH.md #HBM)", "kernels": {}}
def lookup(name):
    """alg key whose base name matches and whose template arguments are a prefix of the kernel's (the kernels gained template
    parameters after this table was written: hconv16_kernel<1, 0, 0> is the forward form hconv16_kernel<1>)"""
    base, _, targs = name.partition("<")
    targs = [t.strip() for t in targs.rstrip(">").split(",")] if targs else []
    for key in alg:
        kb, _, ka = key.partition("<")
        ka = [t.strip() for t in ka.rstrip(">").split(",")] if ka else []
        if kb == base and targs[:len(ka)] == ka:
            return key
    return None


for k in sorted(set(fe) | set(wr)):
    name = lookup(k.split(" grid=")[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip())
    if name is None:
        continue
    f = 2.0 * 1024 * sum(fe.get(k, [0])) / max(1, len(fe.get(k, [])))
    w = 1024.0 * sum(wr.get(k, [0])) / max(1, len(wr.get(k, [])))
    res["kernels"][k] = {"what": alg[name][0], "launches": len(fe.get(k, [])), "hbm_read_bytes": f, "hbm_write_bytes": w,
                         "algorithmic_bytes": alg[name][1], "traffic_over_algorithmic": round((f + w) / alg[name][1], 3)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
