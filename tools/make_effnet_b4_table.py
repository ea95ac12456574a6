"""Key / shape table of the PUBLISHED efficientnet-b4 state_dict (efficientnet_pytorch 0.7.1 naming, as wrapped by smp < 0.5), written
down from the architecture table -- NOT from srbh_amd/encoders.py, which tests/test_model_cpu.py checks against it (a18 anchor:
segmentation_models_pytorch is absent, so output parity cannot be pinned here; structure can).

EfficientNet-B4 = B0 scaled by width 1.4 / depth 1.8 (Tan & Le 2019, table 1 + compound scaling; channel rounding to multiples of 8):
  stem 3x3 s2 -> 48;  7 stages (repeats, kernel, stride, expand, out): (2,3,1,1,24) (4,3,2,6,32) (4,5,2,6,56) (6,3,2,6,112) (6,5,1,6,160)
  (8,5,2,6,272) (2,3,1,6,448);  squeeze-excite width = max(1, int(block_in * 0.25));  head 1x1 -> 1792;  fc 1792 -> 1000.
Self-check: the table must add up to the published parameter count of efficientnet-b4, 19 341 616 (17 548 616 without the classifier,
the figure the reference's author records at mymodels.py:765 as 17.55 M).

    python tools/make_effnet_b4_table.py  ->  tests/golden/efficientnet_b4_keys.json
"""
import json
import os

STEM, HEAD, CLASSES = 48, 1792, 1000
STAGES = [(2, 3, 1, 1, 24), (4, 3, 2, 6, 32), (4, 5, 2, 6, 56), (6, 3, 2, 6, 112), (6, 5, 1, 6, 160), (8, 5, 2, 6, 272), (2, 3, 1, 6, 448)]


def bn(prefix, c, t):
    t[prefix + ".weight"] = [c]
    t[prefix + ".bias"] = [c]
    t[prefix + ".running_mean"] = [c]
    t[prefix + ".running_var"] = [c]
    t[prefix + ".num_batches_tracked"] = []


def table(in_channels=3):
    t = {}
    t["_conv_stem.weight"] = [STEM, in_channels, 3, 3]
    bn("_bn0", STEM, t)
    cin, i = STEM, 0
    strides = []
    for rep, k, s, e, cout in STAGES:
        for r in range(rep):
            p = f"_blocks.{i}."
            mid = cin * e
            if e != 1:
                t[p + "_expand_conv.weight"] = [mid, cin, 1, 1]
                bn(p + "_bn0", mid, t)
            t[p + "_depthwise_conv.weight"] = [mid, 1, k, k]
            bn(p + "_bn1", mid, t)
            sq = max(1, int(cin * 0.25))
            t[p + "_se_reduce.weight"] = [sq, mid, 1, 1]
            t[p + "_se_reduce.bias"] = [sq]
            t[p + "_se_expand.weight"] = [mid, sq, 1, 1]
            t[p + "_se_expand.bias"] = [mid]
            t[p + "_project_conv.weight"] = [cout, mid, 1, 1]
            bn(p + "_bn2", cout, t)
            strides.append(s if r == 0 else 1)
            cin = cout
            i += 1
    t["_conv_head.weight"] = [HEAD, cin, 1, 1]
    bn("_bn1", HEAD, t)
    return t, strides


def nparams(t):
    n = 0
    for k, shape in t.items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        m = 1
        for d in shape:
            m *= d
        n += m
    return n


if __name__ == "__main__":
    t, strides = table(3)
    n = nparams(t)
    assert n + HEAD * CLASSES + CLASSES == 19_341_616, n          # published efficientnet-b4 parameter count
    assert n == 17_548_616                                          # ... without _fc: the reference author's "17.55 M"
    # smp's efficientnet-b4 encoder: features after the stem and after blocks 6, 10, 22, 32 (stage_idxs), channels (3,48,32,56,160,448)
    feats, c, stride_acc, out = [], STEM, 2, {}
    i = 0
    per_block = []
    for rep, k, s, e, cout in STAGES:
        for r in range(rep):
            stride_acc *= (s if r == 0 else 1)
            per_block.append([cout, stride_acc])
            i += 1
    stage_idxs = (6, 10, 22, 32)
    out = {"keys": t, "n_params_without_fc": n, "n_params_published": 19_341_616, "block_strides": strides,
           "stage_idxs": list(stage_idxs), "features_channels_strides": [[3, 1], [STEM, 2]] + [per_block[j - 1] for j in stage_idxs]}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "efficientnet_b4_keys.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(path, len(t), "keys", n, "params")
