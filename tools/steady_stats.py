"""developer aid: per-kernel time of the LAST n steps of a bench run from a rocprofv3 --kernel-trace csv (the first steps
carry MIOpen's solver search, which runs every applicable kernel incl. the naive ones).  A step boundary = a launch of
conv_first_kernel (the RRDBNet forward's first launch, one per step).   usage: steady_stats.py <dir with *kernel_trace.csv> <n_steps> [top] [calls]   (calls: sort by launch count)"""
import csv, glob, sys, collections
d, n = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 30
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
bounds = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]                   # a training step ends with libsrbh's one Adam launch
if len(bounds) < 2:                                                                  # (other workloads: one conv_first per RRDBNet forward -- the pipelined training step has several)
    bounds = [i for i, r in enumerate(rows) if "conv_first_kernel" in r[2]]
start = bounds[-n - 1] if len(bounds) > n else bounds[0]
end = bounds[-1]
sel = rows[start:end]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in sel:
    a = agg[k.replace("(anonymous namespace)::", "")[:110]]
    a[0] += 1; a[1] += e - s
steps = max(1, len([b for b in bounds if start <= b < end]))
tot = sum(v[1] for v in agg.values())
wall = (sel[-1][1] - sel[0][0]) if sel else 0
print("steps %d | kernel time %.2f ms/step | wall %.2f ms/step | %d launches/step" % (steps, tot / steps / 1e6, wall / steps / 1e6, len(sel) // steps))
bycalls = len(sys.argv) > 4 and sys.argv[4] != "--stock"
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0 if bycalls else 1])[:top]:
    print("%7.3f ms/step %5.1f %% %6d calls/step  %s" % (v[1] / steps / 1e6, 100.0 * v[1] / tot, v[0] // steps, k))
GROUPS = (("RRDBNet (trunk, first/tail convs)", ("ptrunk", "ptail", "conv_first", "rrdb", "poison", "upconv", "tail_", "conv3x3_f16")),
          ("encoder / decoders: libsrbh depthwise, 1x1 and decoder 3x3 convs, SE, training BatchNorm, upsample+concat", ("dw_fwd", "dw_bwd", "dw_reduce", "dw_lds", "dconv", "mbconv_mid", "se_train", "se_hidden", "se_gate", "se_bwd", "affine_act", "bn_act_train", "bn_large", "up2_cat", "pw_gemm", "pw_wgrad", "transpose_many", "stem_conv", "bn_eval")),
          ("head (libsrbh hconv / hwgrad / BN / elementwise)", ("hconv", "hwgrad", "hbwd16", "hblock16", "hpack", "bn_", "relu_mask", "nchw_to_nhwc", "nhwc_to_nchw", "ps2_", "add_inplace", "aggregate", "chan_sum", "bias_grad")),
          ("optimizer (libsrbh Adam; multi_tensor_apply)", ("multi_tensor_apply", "adam_kernel")),
          ("tiled prediction: mosaic accumulate / finalize", ("mosaic_",)),
          ("losses (libsrbh)", ("wmse_", "cedice_")))
gs = collections.OrderedDict((g[0], [0, 0]) for g in GROUPS)
gs["encoder / decoders / losses: stock ops (MIOpen, rocBLAS, ATen)"] = [0, 0]
for k, v in agg.items():
    for name, pats in GROUPS:
        if any(p in k for p in pats):
            gs[name][0] += v[0]; gs[name][1] += v[1]
            break
    else:
        gs["encoder / decoders / losses: stock ops (MIOpen, rocBLAS, ATen)"][0] += v[0]; gs["encoder / decoders / losses: stock ops (MIOpen, rocBLAS, ATen)"][1] += v[1]
print("by group:")
for name, v in gs.items():
    print("%7.3f ms/step %5.1f %% %6d launches/step  %s" % (v[1] / steps / 1e6, 100.0 * v[1] / tot, v[0] // steps, name))
if "--stock" in sys.argv:      # every kernel of the stock-op group (what a libsrbh replacement of the decoder convs would remove)
    print("stock-op kernels (all):")
    allp = [p for _, pats in GROUPS for p in pats]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if not any(p in k for p in allp):
            print("%7.3f ms/step %6.1f calls/step  %s" % (v[1] / steps / 1e6, v[0] / steps, k[:150]))
