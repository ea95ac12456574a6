#!/usr/bin/env python3
"""bench.py -- tiles/s of the MI355X-native hot path (BASELINE.json metric) + rooflines + CPU baselines.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Headline workload (default, BASELINE.json configs[1]): RRDBNet x4 (23 RRDB, 64 feat) forward_feature, batch 32
synthetic 64x64 tiles per GPU, inputs resident in HBM, random-init weights of the reference architecture.
A "step" is one forward_feature over one batch.  Tiles are independent, so ranks shard the tile stream with
no data-path collective ("scaling": "weak"); the only collectives are the timing barrier / max.

One JSON line is printed by rank 0 (contract in the task statement).  Extra keys of the default run:
  roofline     -- MFMA roofline of the dominant kernel (persistent trunk), launch duration measured live with HIP events
                  on the launch stream; `traffic` is the PMC figure RECORDED under profiles/ (`traffic_source` says so).
  cpu_baseline -- the CPU oracle (oracle/srbh_oracle.py, proven equal to the reference) timed on this host (N=1 only).
  train_step   -- BASELINE configs[2] (N=1) / configs[3] (N>1): the full training step at batch 64 per GPU, a few timed
                  steps, with per-kernel rooflines of its dominant kernels (trunk: MFMA; head convs: HBM), the gradient
                  all-reduce's isolated and exposed time (N>1) and a CPU baseline at B=4 (N=1).
  predict      -- BASELINE configs[4]: tiled city inference incl. quantise + integer mosaic, the FIRST 30 of the 301 synthetic
                  cities (an unbiased draw of the stated size distribution: 3 cities > 10 k cells, every ragged-tail shape),
                  tiles/s and p50 per-city latency.
  parity       -- the headline output on one tile against the strict-fp32 GPU path and (N=1) the CPU oracle: rel-L2, RMSE.
`--workload train|predict|epoch` run one of those as the headline instead (longer, all cities / a whole epoch).
`--no-extras` drops the two sub-objects from the default run.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

GFLOP_PER_TILE_FEATURE = 146.630   # 2*9*Cin*Cout*H*W over the 350 convs of forward_feature
GFLOP_PER_TILE_PREDICT = 155.65    # + eval head 8.12 + encoder / decoders ~0.9
PEAK_F16_TFLOPS = 2500.0           # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md), never the sparse figure
PEAK_F32_MFMA_TFLOPS = 157.3       # dense fp32 matrix rate
PEAK_HBM_GBS = 8000.0              # HBM3E


def _host_cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def _recorded_traffic(kernel_name):
    """HBM bytes per trunk launch from the newest PMC summary under profiles/ (tools/pmc_traffic.py: separate rocprofv3
    --pmc passes of this very command, FETCH_SIZE doubled per the guide).  A recorded constant, NOT measured in this run:
    a file whose dominant kernel is not the kernel this run launched (`srbh_trunk_kernel_name()`) is REFUSED (-> null),
    so the figure cannot silently outlive the kernel it was measured on."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    for f in reversed(files):
        try:
            rec = json.load(open(f))
            val = rec["dominant_kernel_hbm_bytes_per_launch"]
        except Exception:
            continue
        names = [k.split(" grid=")[0] for k, e in sorted(rec.get("kernels", {}).items(), key=lambda kv: -kv[1].get("launches", 0))
                 if "ptrunk" in k and "reset" not in k]
        rec_name = rec.get("dominant_kernel", names[0] if names else None)
        if rec_name is None or kernel_name is None or rec_name.split("<")[0] != kernel_name.split("<")[0]:
            return None, f"profiles/{os.path.basename(f)} REFUSED: recorded on '{rec_name}', this run launched '{kernel_name}'"
        return val, "profiles/" + os.path.basename(f)
    return None, None


def cpu_baseline(sd, seconds_budget=20.0, want_ref=None):
    """Oracle forward_feature on the host cores, bounded to ~seconds_budget of wall time.
    `want_ref` (a list): the oracle's output for the first tile of the sample is appended to it -- the `parity` field of the
    headline compares the HIP path's output on the same tile with it (the oracle runs in this leg only).
    oneDNN scales badly past a few dozen threads on these small convs, so the thread count is capped at 32
    (`cores` reports the threads actually used).  Protocol: one timed B=1 call decides the sample size."""
    from oracle import srbh_oracle as O  # the checker: cpu_baseline leg only
    from srbh_amd import synth
    cores = min(_host_cores(), 32)
    torch.set_num_threads(cores)
    x = synth.tiles(4, 8, 64, seed=1)[:, :3].contiguous()
    sd_cpu = {k: v.float() for k, v in sd.items()}
    t0 = time.perf_counter()
    y_ref = O.rrdbnet_forward_feature(sd_cpu, x[:1])  # first call also creates the oneDNN primitives
    t_first = time.perf_counter() - t0
    if want_ref is not None:
        want_ref.extend([x[:1], y_ref])
    if t_first > seconds_budget / 3:          # very slow host: the single B=1 call is the sample
        return {"value": round(1 / t_first, 4), "unit": "tiles/s", "cores": cores, "kind": "port",
                "sample": f"oracle RRDBNet.forward_feature fp32, ONE B=1 call incl. warm-up ({t_first:.1f} s of CPU work)"}
    bs = 4 if t_first * 4 * 3 < seconds_budget else 1
    times, t_start = [], time.perf_counter()
    # at least 3 calls; keep sampling until ~10 s of CPU work (bounded by the budget) so fast hosts are not under-sampled
    while len(times) < 3 or (time.perf_counter() - t_start < min(10.0, seconds_budget) and len(times) < 64):
        if len(times) >= 3 and time.perf_counter() - t_start >= seconds_budget:
            break
        t0 = time.perf_counter()
        O.rrdbnet_forward_feature(sd_cpu, x[:bs])
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(bs / med, 3), "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"oracle RRDBNet.forward_feature fp32, B={bs} tiles (64x64x3 -> 64x256x256), median of {len(times)} "
                      f"calls after 1 warm-up ({sum(times) + t_first:.1f} s of CPU work)"}


def cpu_baseline_train(rrdb_sd, seconds_budget=30.0):
    """the full training step at B=4 on the host cores (oracle/train_oracle.py: oracle RRDBNet forward under
    no_grad + stock-op encoder/decoders + oracle head forward/backward + oracle losses + Adam), bounded to
    ~seconds_budget: one warm-up step, then up to 3 timed steps."""
    from oracle import srbh_oracle as O
    from oracle.train_oracle import CpuTrainStep
    from srbh_amd.harness import synthetic_batch
    from srbh_amd.models import SRRegress_Cls_feature
    cores = min(_host_cores(), 32)
    torch.set_num_threads(cores)
    torch.manual_seed(1337)
    model = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True,
                                  chans_build=7)
    ts = CpuTrainStep(rrdb_sd, model)
    B = 4
    batch = synthetic_batch(B, 1337, "cpu", aggregate=O.aggregate_torch)
    t0 = time.perf_counter()
    ts(batch)
    t_first = time.perf_counter() - t0
    times = []
    while len(times) < 3 and (not times or time.perf_counter() - t0 + times[-1] < seconds_budget):
        t1 = time.perf_counter()
        ts(batch)
        times.append(time.perf_counter() - t1)
    med = sorted(times)[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"oracle training step fp32 at B={B} (RRDBNet fwd no-grad + encoder/decoders/head fwd+bwd + losses + Adam), "
                      f"median of {len(times)} steps after 1 warm-up ({sum(times) + t_first:.1f} s of CPU work)"}


def _timed(fn, n, dev):
    """average duration (ms) of fn() over n calls, HIP events on the current stream (= the stream libsrbh launches on)."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def head_kernel_rooflines(dev, B):
    """The training step's dominant HEAD kernels timed on their own at the step's batch size (16-channel 256x256 maps:
    SURVEY 8d bounds the head by HBM).  Algorithmic bytes per launch: fp32 NHWC input + output (+ both operands for the
    weight gradient)."""
    from srbh_amd import hrfuse as H, hrfuse_autograd as HA
    px = B * 256 * 256
    out = []
    conv = torch.nn.Conv2d(16, 16, 3, 1, 1, bias=False).to(dev)
    x = H.to_nhwc(torch.randn(B, 16, 256, 256, device=dev))
    g = H.to_nhwc(torch.randn(B, 16, 256, 256, device=dev))
    pk, pg = H._PackedConv(), HA._PackedGrad()
    h16 = H.head_h16()
    for name, fn, nbytes in (
            (f"{'hconv16_kernel (fp16' if h16 else 'hconv_f32_kernel (fp32'} operands) 16->16 3x3 fwd + BN statistics", lambda: H.hconv([x], conv, pk, want_stats=True), px * (64 + 64)),
            (f"{'hconv16_kernel (bf16' if h16 else 'hconv_f32_kernel (fp32'} operands) 16->16 3x3 data gradient", lambda: HA.conv_dgrad(g, conv.weight, pg), px * (64 + 64)),
            (f"{'hwgrad16_kernel (bf16' if h16 else 'hwgrad_f32_kernel (fp32'} operands) 16->16 3x3 weight gradient (+ its 2 reduce launches)", lambda: HA.conv_wgrad([x], None, g, 16, 3), px * (64 + 64))):
        ms = _timed(fn, 10, dev)
        out.append({"kernel": f"{name} @256x256, B={B}", "bound": "hbm", "avg_launch_ms": round(ms, 4),
                    "algorithmic_bytes_per_launch": nbytes, "achieved": round(nbytes / ms / 1e6, 1), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4)})
    return out


def _make_nets(args, dev, isaggre):
    from srbh_amd import synth
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    sd = synth.rrdbnet_state_dict(num_block=args.num_block, seed=1337, mode="init")
    net_hr = RRDBNet(3, 3, num_block=args.num_block)
    net_hr.load_state_dict(sd)
    torch.manual_seed(1337)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=isaggre,
                                chans_build=7)
    return sd, net_hr.to(dev), net.to(dev)


_DETAILS = {}          # full per-call tables of the run (written by --details; the printed line carries the top rows only)


def _dist_info(dist):
    """what the process group actually is: proves which backend carried the collectives and how many ranks it saw"""
    if dist is None:
        return {"backend": None, "world_size_seen": 1, "rccl_version": None}
    ver = None
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    return {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "rccl_version": ver}


def _max_over_ranks(vals, dev, dist):
    if dist is None:
        return vals
    t = torch.tensor(vals, dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def _pipe_images(dev):
    from srbh_amd.harness import pipe_images
    return pipe_images(dev)


def bench_train(args, rank, world, dev, dist, steps, warmup, batch=64, epoch_tiles=None, with_cpu=True, with_kernels=True):
    """BASELINE configs[2]/[3]: RRDBNet forward (no grad) + SRRegress_Cls_feature forward/backward + Adam, `batch` tiles
    per GPU, gradients averaged over ranks by bucketed RCCL all-reduces launched from autograd hooks (overlapped with
    backward).  `epoch_tiles`: run one data-parallel pass over that many synthetic tiles (batches drawn on the device)
    instead of `steps` repeats of one fixed batch."""
    from srbh_amd.harness import TrainStep, synthetic_batch, train_epoch
    sd, net_hr, net = _make_nets(args, dev, True)
    sync_bn = os.environ.get("SRBH_SYNC_BN", "0") == "1"        # default: per-rank BatchNorm statistics
    # SRBH_TRAIN_GRAPH=1 (fixed batch): the step replayed as ONE HIP graph.  On one GPU measured equal to eager launches
    # (53.0 ms both): with the fused optimizer the host keeps up, the step is bound by its ~1 300 kernels and the gaps between them.
    # N > 1: the graph ends with backward, the all-reduce buckets + Adam follow eagerly (no overlap with backward) -- the switch for
    # a node whose host cannot issue N eager steps at once (harness.TrainStep).
    use_graph = os.environ.get("SRBH_TRAIN_GRAPH", "0") == "1" and not epoch_tiles and not (world > 1 and sync_bn)
    ts = TrainStep(net_hr, net, dev, world=world, sync_bn=sync_bn, timing=True, status_every=0, graph=use_graph)
    fixed = synthetic_batch(batch, 1337 + rank, dev)
    # the pipelined step (harness.TrainStep): the RRDBNet features of the NEXT batch are computed on a second stream beside this step's
    # small-kernel phases.  Every timed step still contains one full trunk pass (that of the batch the following step consumes; the last
    # one is computed and never used), and the final synchronize covers the second stream.  SRBH_TRAIN_PIPELINE=0: the serial step.
    pipe = os.environ.get("SRBH_TRAIN_PIPELINE", "1") == "1" and not use_graph
    # the timed steps alternate between TWO batches (round-5 VERDICT, hygiene): the announced next batch is then a different tensor from the
    # running one, as in an epoch (graph mode replays one static batch)
    other = fixed if use_graph else synthetic_batch(batch, 7331 + rank, dev)
    pair = (fixed, other)
    nw = max(warmup, 5 if use_graph else 2)
    for i in range(nw):          # (world > 1: step 1 records the bucket plan; graph: 3 eager steps, then the capture; pipeline: step 1 fills it)
        ts(pair[i % 2], next_batch=pair[(i + 1) % 2] if pipe else None)
    if use_graph:
        fixed = ts.static_batch()
        pair = (fixed, fixed)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    exposed = []
    t0 = time.perf_counter()
    p0 = ts.pipelined_steps
    if epoch_tiles:
        steps, tiles, loss = train_epoch(ts, epoch_tiles, batch, rank, world, dev)
    else:
        for i in range(nw, nw + steps):
            loss, _ = ts(pair[i % 2], next_batch=pair[(i + 1) % 2] if pipe else None)
            if ts.reducer is not None:
                exposed.append(ts.reducer._last_events)
        tiles = batch * world * steps
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    (elapsed,) = _max_over_ranks([elapsed], dev, dist)
    net_hr.check_status()
    # how many of the timed steps consumed prefetched features, on the slowest and the fastest rank (a rank whose prefetch never hits
    # runs the serial step and drags the job: must be visible in an N-GPU line)
    hits = ts.pipelined_steps - p0
    hit_lo, hit_hi = hits, hits
    if dist is not None:
        hh = torch.tensor([float(hits), -float(hits)], dtype=torch.float64, device=dev)
        dist.all_reduce(hh, op=dist.ReduceOp.MAX)
        hit_hi, hit_lo = int(hh[0]), int(-hh[1])
    comm = None
    if ts.reducer is not None:
        ex = [a.elapsed_time(b) for a, b in exposed] or [0.0]
        iso = ts.reducer.isolated_comm_ms()
        ex_max, iso_max = _max_over_ranks([sum(ex) / len(ex), iso], dev, dist)
        comm = {"prefetch_hits_per_rank": {"min": hit_lo, "max": hit_hi, "of_steps": steps}, "buckets": ts.reducer.n_buckets, "grad_bytes": int(sum(b["flat"].numel() * 4 for b in ts.reducer.plan)),
                "comm_ms": round(iso_max, 3), "exposed_comm_ms": round(ex_max, 3),
                "note": "comm_ms = the step's bucketed all-reduces on their own; exposed_comm_ms = device time between the "
                        "end of backward and the last bucket landing (max over ranks)"}
    strict_ms = None
    if with_kernels and not epoch_tiles and world == 1:
        # the same step with the head's convolutions on the exact-fp32 matrix cores (the mode every <= 5e-5 gradient-parity test
        # pins): reported NEXT to the mixed-precision headline so that nobody reads 1e-3-outputs / 3e-2-gradients as fp32 parity
        _, net_hr2, net2 = _make_nets(args, dev, True)
        ts2 = TrainStep(net_hr2, net2, dev, world=1, status_every=0, head_precision="f32")
        for _ in range(3):
            ts2(fixed, next_batch=fixed if pipe else None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            ts2(fixed, next_batch=fixed if pipe else None)
        torch.cuda.synchronize()
        strict_ms = (time.perf_counter() - t1) / 3 * 1e3
        del ts2, net_hr2, net2
        torch.cuda.empty_cache()
    if rank != 0:
        return None
    ms_step = elapsed / steps * 1e3
    gf_tile = 146.63 + 8.12 * 3 + 0.9 * 3      # RRDB fwd + head fwd/bwd (~3x fwd) + encoder/decoders
    line = {
        "metric": "tiles/sec (64x64x8ch->256x256 height) fwd+bwd", "value": round(tiles / elapsed, 2), "unit": "tiles/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": f"f16 operands/f32 acc (RRDB), head convs: {ts.head_precision} (f16 = forward fp16 operands, data and weight gradients bf16 operands, fp32 accumulation; BN + losses + Adam fp32)",
        "data": "synthetic" + (" (batches drawn on the device each step)" if epoch_tiles else (" (one static batch: graph replay)" if use_graph else " (two alternating batches)")),
        "config": {"workload": (f"one data-parallel pass over {epoch_tiles} synthetic train tiles (BASELINE.json configs[3]), " if epoch_tiles else "")
                               + f"full train step: RRDBNet fwd (no grad) + SRRegress_Cls_feature fwd/bwd + Adam, batch {batch}/GPU "
                               "(BASELINE.json configs[2])", "batch": batch, "global_batch": batch * world,
                   "parallelism": (f"dp{world} (fwd+bwd replayed as one HIP graph, then the bucketed RCCL grad all-reduce + Adam)" if use_graph else
                                   f"dp{world} (bucketed RCCL grad all-reduce launched from autograd hooks)") if world > 1 else
                   ("1 GPU, the whole step replayed as one HIP graph" if use_graph else "1 GPU, eager launches"),
                   "pipeline": (f"RRDBNet features of batch k+1 on a second stream beside step k's encoder / decoder phases, {_pipe_images(dev)} images "
                                f"per trunk launch ({ts.pipelined_steps - p0} of the {steps} timed steps consumed prefetched features; "
                                "one full trunk pass inside every timed step)") if pipe and ts.pipelined_steps else "none (serial step)"},
        "whole_step": {"gflop_per_tile": gf_tile, "achieved_tflops": round(gf_tile * tiles / elapsed / 1e3, 2),
                       "note": "mixes the MFMA-bound trunk with the HBM-bound head: not a roofline, see `kernels`"},
        "final_loss": float(loss)}
    if strict_ms is not None:
        line["strict_f32_head"] = {"ms_per_step": round(strict_ms, 3), "value": round(batch / strict_ms * 1e3, 2), "unit": "tiles/s",
                                   "note": "same step, head convolutions + their gradients on exact-fp32 matrix cores (head_precision='f32': "
                                           "the mode the <=5e-5 gradient-parity tests pin); the headline's 'f16' mode keeps outputs <= 1e-3 but "
                                           "its parameter gradients are compared with this mode's in `parity` (heads <= 4e-3, upstream groups direction-accurate)"}
    if comm:
        comm.update(_dist_info(dist))
        line["comm"] = comm
    if with_kernels:
        # per-kernel rooflines: the trunk launch inside THIS step (HIP events via libsrbh's hook), the head kernels on their own
        import ctypes
        from srbh_amd import _lib
        L = _lib.lib()
        ks = []
        acc = 0.0
        if L.srbh_trunk_timing(1) == 0:
            try:
                for _ in range(3):
                    with torch.no_grad():
                        net_hr.forward_feature(fixed[0][:, :3])
                    ms = ctypes.c_float(0.0)
                    _lib.check(L.srbh_trunk_last_ms(ctypes.byref(ms)), "srbh_trunk_last_ms")
                    acc += ms.value
            except RuntimeError:      # SRBH_PERSISTENT=0 (per-layer launches: no single trunk kernel to time) -> no trunk row
                acc = 0.0
            L.srbh_trunk_timing(0)
        if acc > 0.0:
            tg = args.num_block * 3 * 4096 * 18 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64) / 1e9 * batch
            ks.append({"kernel": f"persistent trunk (345 dense-block convs), B={batch}", "bound": "mfma", "avg_launch_ms": round(acc / 3, 4),
                       "algorithmic_gflop_per_launch": round(tg, 1), "achieved": round(tg / (acc / 3), 2), "peak": PEAK_F16_TFLOPS,
                       "unit": "TFLOP/s", "frac": round(tg / (acc / 3) / PEAK_F16_TFLOPS, 4)})
        from srbh_amd import hrfuse as _H
        with _H.head_precision(ts.head_precision):
            ks += head_kernel_rooflines(dev, batch)
        line["kernels"] = ks
    if with_kernels and world == 1:       # (N > 1: the other ranks have returned -- a further step's all-reduce would have no partner)
        # the WHOLE head (+ losses) of this very step against the HBM roofline: every libsrbh call of two extra steps bracketed by
        # HIP events and priced with the algorithmic bytes its arguments imply (srbh_amd/kprof.py); outside the timed region
        from srbh_amd.kprof import KernelProfile
        _lib.path_counters(reset=True)
        with KernelProfile() as kp:
            for _ in range(2):
                ts(fixed)
        rows, totals = kp.table(steps=2)
        # which FORM of the head entry points ran in those 2 steps (persistent kernel vs template, fused vs split): no silent fallback
        line["head_paths_2steps"] = _lib.path_counters()
        top = rows[:8]
        totals["note"] = ("every libsrbh head / loss call of the step (HIP events around each call); `kernels` = the 8 largest "
                          "(all rows: --details / profiles/); bytes = each tensor once at its stored element size; peak 8000 GB/s")
        totals["covered_ms_per_step"] = round(sum(r["ms_per_step"] for r in top), 3)
        totals["kernels"] = top
        _DETAILS["head_roofline_rows"] = rows
        line["head_roofline"] = totals
        # the encoder / decoder calls of the same step (libsrbh entry points only: the decoders' 3x3 convs are MIOpen's), same method
        from srbh_amd import encoders as _E0
        _E0.stock_ops_reset()
        with KernelProfile(group="encdec") as kp2:
            for _ in range(2):
                ts(fixed)
        rows2, tot2 = kp2.table(steps=2)
        agg = {}
        for r in rows2:                                   # per entry point (the per-shape rows are ~300: one per layer and direction)
            a = agg.setdefault(r["kernel"].split(" ")[0], [0.0, 0.0, 0.0])
            a[0] += r["calls_per_step"]
            a[1] += r["ms_per_step"]
            a[2] += r["algorithmic_MB_per_call"] * r["calls_per_step"]
        from srbh_amd import encoders as _E
        tot2["stock_ops"] = _E.stock_ops_summary()
        tot2["stock_ops"]["note"] = ("encoder / decoder call sites that ran stock PyTorch-ROCm ops (MIOpen / ATen) in the 2 profiled steps, "
                                     "counted per (site, shape) in encoders.STOCK_OPS: no silent fallback")
        by_entry = [{"entry": k, "calls_per_step": round(v[0], 1), "ms_per_step": round(v[1], 3), "us_per_call": round(v[1] / max(v[0], 1e-9) * 1e3, 1),
                                   "algorithmic_MB_per_step": round(v[2], 1)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])]
        tot2["by_entry_point"] = by_entry[:5]
        _DETAILS["encdec_by_entry_point"] = by_entry
        _DETAILS["encdec_rows"] = rows2
        tot2["note"] = ("libsrbh encoder / decoder calls of the step, HIP events around each call; top 5 entry points (all: --details); "
                        "planes are 2x2 .. 64x64: latency chains, the HBM fraction is not their roofline (DESIGN.md 3.10)")
        line["encdec_kernels"] = tot2
    if with_kernels and world == 1 and not epoch_tiles:
        # the mixed mode's parameter gradients against the exact-fp32 graph of the SAME step (weights as they stand after the timed steps, the
        # fixed batch, lr 0): measured in this run, with the stated per-group tolerance (srbh_amd/gradcheck.py, tests/test_gpu_grad_parity.py)
        from srbh_amd import gradcheck
        try:
            line["parity"] = gradcheck.mixed_vs_exact(net_hr, net, fixed, dev)
        except Exception as e:
            line["parity"] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if with_cpu and world == 1:
        line["cpu_baseline"] = cpu_baseline_train(sd)
    return line


def bench_predict(args, rank, world, dev, dist, n_cities, warmup, batch=256, small=False):
    """BASELINE configs[4]: urban-centre tiled inference.  A "step" is ONE synthetic city: its grid cells (64x64x8 LR tiles,
    stride 48 as in the reference's grid loader) are sharded over the ranks, each rank runs RRDBNet features -> eval head ->
    quantise + integer mosaic on the device, the integer mosaics are summed over ranks (bit-identical to the serial
    result) and finalised (argmax / normalise).  City sizes: 301 cell counts drawn log-uniform in [200, 20000] (numpy
    default_rng(2024)); no per-city grid counts ship with the reference.  `small`: the n_cities cities closest
    to the median size (the default run's bounded sample) instead of the first n_cities.  GeoTIFF IO is outside the path."""
    import numpy as np
    from srbh_amd.harness import predict_tiles
    from srbh_amd.mosaic import Mosaic
    _, net_hr, model = _make_nets(args, dev, False)
    net_hr.eval()
    model.eval()
    counts = np.exp(np.random.default_rng(2024).uniform(np.log(200.0), np.log(20000.0), 301)).astype(int)
    # MIOpen solver search for the stock-op encoder / decoders: without it a few forward convolutions run on MIOpen's naive
    # kernels (immediate-mode fallback).  One-time cost in the warm-up; every batch has one of four shapes.  (Not used
    # for the training workload: the backward searches take minutes.)
    torch.backends.cudnn.benchmark = os.environ.get("SRBH_MIOPEN_FIND", "1") == "1"
    if small:
        order = np.argsort(np.abs(np.log(counts) - np.log(np.median(counts))))
        todo = [int(counts[i]) for i in sorted(order[:n_cities])]
    else:
        todo = [int(c) for c in counts[:n_cities]]
    pad_to = 32 if batch % 32 == 0 and batch > 32 else None          # ragged tails run as 32/64/96/... tiles, not the full batch
    n_warm = min(warmup, 2)
    warm = [min(int(c), 256) for c in counts[-n_warm:]] if n_warm else []
    if pad_to:
        # every padded tail shape once on every rank (MIOpen searches per tensor shape; outside the timed region)
        with torch.no_grad():
            xw = torch.randn((batch, 8, 64, 64), device=dev) * 0.25 + 0.35
            for q in range(pad_to, batch + 1, pad_to):
                model(xw[:q], net_hr.forward_feature(xw[:q, :3]))
        torch.cuda.synchronize()

    def city(n, seed):
        gw = int(np.ceil(np.sqrt(n)))
        gh = (n + gw - 1) // gw
        pos = [[(i % gw) * 48, (i // gw) * 48, 64, 64] for i in range(n)]
        g = torch.Generator(device=dev).manual_seed(seed)        # inference tiles: N(0.35, 0.25), not clipped
        tiles = torch.randn((n, 8, 64, 64), generator=g, device=dev) * 0.25 + 0.35
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        m = Mosaic((gh * 48 + 16) * 4, (gw * 48 + 16) * 4, 7, dev)
        predict_tiles(net_hr, model, tiles, pos, m, batch=batch, rank=rank, world=world, pad_to=pad_to)
        if dist is not None:
            m.reduce_to_(dist, dst=0)      # each rank ships only the row band it wrote; rank 0 holds the city
        out = m.finalize() if rank == 0 else None
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        (dt,) = _max_over_ranks([dt], dev, dist)
        del m, out, tiles
        return dt

    for i, n in enumerate(warm):
        city(n, 99 + i)
    from srbh_amd import _lib as _L
    _L.path_counters(reset=True)
    lat = [city(n, 2024 + i) for i, n in enumerate(todo)]
    head_paths = _L.path_counters()          # (graph replays launch without passing the C entry points: these are the eager tail batches + captures)
    torch.backends.cudnn.benchmark = False
    # where a full batch spends its time (rank 0, outside the timed region): each HIP graph of harness._PredictGraph alone, and the batch as
    # predict_tiles schedules it -- the sum of the parts against the whole says what the second stream hides (nothing)
    parts = None
    pg = model.__dict__.get("_srbh_predict_graph")
    if rank == 0 and pg is not None and getattr(pg, "ahead", False):
        def _t(fn, n=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / n * 1e3, 3)
        with torch.no_grad():
            xb = torch.randn((batch, 8, 64, 64), device=dev) * 0.25 + 0.35
            parts = {"trunk_first_tail_convs_graph": _t(pg.g_trunk.replay), "hrfeature_graph": _t(pg.g_hrfeat.replay),
                     "encoder_decoders_graph": _t(pg.g_lr2[0].replay), "reg_seg_graph": _t(pg.g_fuse2[0].replay)}
            parts["sum_of_parts"] = round(sum(parts.values()), 3)
            parts["batch_as_scheduled"] = _t(lambda: pg(xb, batch, xb))
            pg.reset()
            parts["unit"] = f"ms per batch of {batch} tiles, graphs only (no mosaic, no ragged tail)"
    if rank != 0:
        return None
    total, elapsed = sum(todo), sum(lat)
    order = sorted(range(len(lat)), key=lambda i: lat[i])
    mid = order[len(order) // 2]
    return {
        "metric": "tiles/sec (64x64x8ch->256x256 height) tiled inference incl. quantise + mosaic", "value": round(total / elapsed, 2),
        "unit": "tiles/s", "n_gpus": world, "steps": len(todo), "warmup": warmup, "ms_per_step": round(elapsed / len(todo) * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16 operands/f32 acc (RRDB and head convs), f32 BN-affine / stock-op encoder+decoders, integer mosaic", "data": "synthetic cities, random-init weights",
        "config": {"workload": f"sliding-window predict path, {len(todo)} of 301 synthetic cities (cells log-uniform 200..20000, "
                               f"seed 2024{', the ones closest to the median size' if small else ''}), batch {batch}/GPU (BASELINE.json configs[4])",
                   "cities": len(todo), "tiles": total, "parallelism": f"each city's cells sharded x{world}, integer mosaic row bands sent to rank 0"},
        "p50_city_latency_ms": round(lat[mid] * 1e3, 2), "p50_city_tiles": todo[mid],
        "p95_city_latency_ms": round(lat[order[min(len(order) - 1, int(0.95 * len(order)))]] * 1e3, 2),
        "max_city_latency_ms": round(max(lat) * 1e3, 2), "max_city_tiles": max(todo),
        "large_cities": {"n_over_10k_cells": sum(1 for c in todo if c > 10000),
                         "tiles_per_s": round(sum(c for c in todo if c > 10000) / max(1e-9, sum(l for c, l in zip(todo, lat) if c > 10000)), 2)
                         if any(c > 10000 for c in todo) else None},
        "tail_shapes_run": sorted({(c % batch + 31) // 32 * 32 for c in todo if c % batch}) if pad_to else None,
        "roofline": {"bound": "mfma", "achieved": round(total / elapsed / world * GFLOP_PER_TILE_PREDICT / 1e3, 2), "peak": PEAK_F16_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(total / elapsed / world * GFLOP_PER_TILE_PREDICT / 1e3 / PEAK_F16_TFLOPS, 4),
                     "gflop_per_tile": GFLOP_PER_TILE_PREDICT, "trunk_share_of_flops": round(135.44 / GFLOP_PER_TILE_PREDICT, 3),
                     "note": "whole predict path PER GPU (RRDBNet forward_feature 146.63 + eval head 8.12 + encoder/decoders 0.9 GFLOP per tile, "
                             "SURVEY 8d) incl. tile generation hand-off, quantise and mosaic: not one kernel; the trunk kernel's own fraction is the headline's `roofline`"},
        "dist": _dist_info(dist),
        "batch_parts_ms": parts,
        "head_paths_eager_calls": {k: v for k, v in head_paths.items() if v}}


def bench_sr_train(args, rank, world, dev, dist, steps, warmup, batch=8):
    """SURVEY 8f-4 (SR-stage fine-tuning, reference SR/rrdbnet_arch.py:538-592): forward + backward of the GENERATOR -- RRDBNet.forward
    on `batch` 64x64 tiles with a recorded graph, every parameter gradient -- in the training path's precision modes.  A "step" is one
    forward + backward (no optimizer, no discriminator: those are stock ops).  Work: 3 x 146.857 GFLOP per tile (forward, data
    gradients, weight gradients: SURVEY 8d's hook-counted forward figure)."""
    from srbh_amd import rrdbnet_autograd as RA
    from srbh_amd import synth
    from srbh_amd.rrdbnet import RRDBNet
    sd = synth.rrdbnet_state_dict(num_block=args.num_block, seed=1337, mode="init")
    gf_tile = 3 * (146.857 if args.num_block == 23 else (args.num_block * 5.8886 + 11.42))
    out = {}
    modes = [m for m in os.environ.get("SRBH_SR_BENCH_MODES", "fast,mixed,f32").split(",") if m]
    for mode in modes:
        RA.set_train_precision(mode)
        net = RRDBNet(3, 3, num_block=args.num_block)
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).train().enable_training_path(True)
        x = synth.tiles(batch, 8, 64, seed=1337 + rank)[:, :3].contiguous().to(dev)
        w = None
        n_w, n_s = (warmup, steps) if mode != "f32" else (1, max(1, min(steps, 3)))

        def step():
            nonlocal w
            for p in net.parameters():
                p.grad = None
            y = net(x)
            if w is None:
                w = torch.randn(y.shape, generator=torch.Generator(device=dev).manual_seed(4242), device=dev)      # (the same cotangent in every mode)
            (y * w).sum().backward()

        for _ in range(n_w):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n_s):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        (el,) = _max_over_ranks([time.perf_counter() - t0], dev, dist)
        ms = el / n_s * 1e3
        gn = float(sum(p.grad.double().pow(2).sum() for p in net.parameters() if p.grad is not None).sqrt())
        out[mode] = {"ms_per_step": round(ms, 3), "tiles_per_s": round(batch * world / ms * 1e3, 2), "achieved_tflops": round(gf_tile * batch / ms, 2),
                     "steps": n_s, "grad_norm": gn}
        if mode == "fast" and rank == 0:
            out[mode]["kernels"] = _sr_trunk_kernels(net, batch, dev, args.num_block)
            if os.environ.get("SRBH_SR_BENCH_ITERATION", "1") != "0":      # (the stock discriminator: MIOpen searches its kernels on first use, ~1 min)
                out[mode]["iteration"] = _sr_iteration(batch, dev, args.num_block, max(3, min(steps, 8)))
        del net
        torch.cuda.empty_cache()
    RA.set_train_precision("f32")
    if rank != 0:
        return None
    head = out[modes[0]]
    return {"metric": "tiles/sec (64x64 -> 256x256) generator fwd+bwd, SR-stage fine-tuning", "value": head["tiles_per_s"], "unit": "tiles/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fast": "f16 forward / bf16 gradient operands on the trunk's 32x32x16 MFMA kernels (dense blocks), f32 accumulate + residual streams",
                                           "mixed": "f16 / bf16 operands on the head's 16x16x16 kernels", "f32": "exact fp32 matrix cores"}[modes[0]],
            "data": "synthetic tiles, random-init weights",
            "config": {"workload": f"RRDBNet x4 ({args.num_block} RRDB) forward + backward (all parameter gradients), batch {batch}/GPU (SURVEY 8f-4)", "batch": batch, "mode": modes[0],
                       "trunk_forward_calls": dict(_sr_paths()[0]), "trunk_backward_calls": dict(_sr_paths()[1])},      # 'fast' mode: persistent (one launch of the inference trunk's kernel) vs per_layer: no silent fallback
            "roofline": {"bound": "mfma", "achieved": head["achieved_tflops"], "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(head["achieved_tflops"] / PEAK_F16_TFLOPS, 4), "gflop_per_step": round(gf_tile * batch, 1),
                         "note": "whole fwd+bwd (345 dense-block convs x 3 + the non-trunk convs in 'mixed'), not one kernel",
                         "kernels": head.get("kernels")},
            "iteration": head.get("iteration"),
            "modes": out}


def _sr_iteration(batch, dev, num_block, iters):
    """one WHOLE iteration of the reference's SR-stage trainer (SR/rrdbnet_arch.py:538-592: generator forward, pixel + GAN losses, generator backward +
    Adam, discriminator forward / backward on real and fake + Adam, EMA) through RealESRGAN(is_train=True).optimize_parameters(): the generator on
    libsrbh in the mode being measured, discriminator / USM sharpener / losses on stock device ops, every weight pack rebuilt each iteration (the
    weights move)."""
    from srbh_amd.rrdbnet import RealESRGAN
    torch.manual_seed(3)
    m = RealESRGAN(3, 3, num_block=num_block, device=dev, is_train=True)
    g = torch.Generator().manual_seed(9)
    gt = torch.nn.functional.interpolate(torch.rand((batch, 3, 32, 32), generator=g), scale_factor=8, mode="bilinear").to(dev)      # smooth 256 x 256 targets
    lq = torch.nn.functional.avg_pool2d(gt, 4)                                                                                       # 64 x 64
    ld = None
    for _ in range(2):
        m.feed_data({"lq": lq, "gt": gt})
        ld = m.optimize_parameters()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        m.feed_data({"lq": lq, "gt": gt})
        ld = m.optimize_parameters()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"what": "RealESRGAN.optimize_parameters(): generator + discriminator forward / backward, both Adam steps, EMA (no perceptual plug-in)",
            "ms_per_iteration": round(ms, 3), "tiles_per_s": round(batch / ms * 1e3, 2), "iterations": iters, "l_g_pix": float(ld["l_g_pix"])}


def _sr_trunk_kernels(net, batch, dev, num_block, reps=10):
    """the three launches that carry 92 % of the generator step's FLOPs, timed alone with events on the current stream (the stream they are issued
    on): the persistent trunk forward, its bf16 form for the data gradients, and all dense blocks' weight + bias gradients (+ the ordered reduce).
    Each does 23 x 3 x 26 624 x 9 x 2 FLOP per pixel = 135.44 GFLOP per 64 x 64 tile."""
    import ctypes as C
    from srbh_amd import _lib
    from srbh_amd import rrdbnet_autograd as RA
    L = _lib.lib()
    feat = torch.randn((batch, 64, 64, 64), device=dev) * 0.5
    g = torch.randn((batch, 64, 64, 64), device=dev) * 1e-3
    gf = 2.0 * 9 * 26624 * num_block * 3 * batch * 64 * 64 / 1e9

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    lease = [None]

    def fwd():
        if lease[0] is not None:
            lease[0].release()
        _, lease[0] = RA._trunk_fast_forward(net, feat)

    t_f = timed(fwd)
    if not RA.TRUNK_FWD_PATHS["persistent"]:
        return None
    ws = lease[0].ws
    t_b = timed(lambda: RA._trunk_fast_backward(net, lease[0], g, {}))
    rows = [{"kernel": "ptrunk3_kernel<0, 0>: forward of the 345 dense-block convs, one launch (+ the fp16 conversion of its input)", "avg_ms": round(t_f, 4)}]
    if ws.get("twg") is not None and RA.TRUNK_BWD_PATHS["persistent"] and RA.TRUNK_WGRAD:
        dw = torch.empty(num_block * 3 * 9 * 26624, device=dev)
        db = torch.empty(num_block * 3 * 192, device=dev)
        t_w = timed(lambda: _lib.check(L.srbh_trunk_wgrad(num_block, ws["D"].data_ptr(), ws["nb"], ws["G"].data_ptr(), ws["nb"], batch, 64, 64, dw.data_ptr(),
                                                          db.data_ptr(), ws["twg"].data_ptr(), _lib.stream_ptr()), "trunk_wgrad"))
        rows.append({"kernel": "ptrunk3_kernel<0, 1>: the 345 data-gradient convs, one launch (+ scaling, conversion, packs cached)", "avg_ms": round(t_b - t_w, 4)})
        rows.append({"kernel": "trunk_wgrad_kernel + trunk_wgrad_reduce_kernel: weight and bias gradients of all dense blocks", "avg_ms": round(t_w, 4)})
    else:
        rows.append({"kernel": "trunk backward (data + weight gradients)", "avg_ms": round(t_b, 4), "gflop": 2 * gf})
    lease[0].release()
    for r in rows:
        f = r.pop("gflop", gf)
        r.update({"algorithmic_gflop": round(f, 1), "achieved": round(f / r["avg_ms"], 1), "unit": "TFLOP/s", "frac": round(f / r["avg_ms"] / PEAK_F16_TFLOPS, 4)})
    return rows


def _sr_paths():
    from srbh_amd import rrdbnet_autograd as RA
    return RA.TRUNK_FWD_PATHS, RA.TRUNK_BWD_PATHS


def bench_feature(args, rank, world, dev, dist):
    from srbh_amd import synth
    from srbh_amd.rrdbnet import RRDBNet
    sd = synth.rrdbnet_state_dict(num_block=args.num_block, seed=1337, mode="init")
    net = RRDBNet(3, 3, num_block=args.num_block)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    B = args.batch
    x = synth.tiles(B, 8, 64, seed=1337 + rank)[:, :3].contiguous().to(dev)  # rgbseq=[0,1,2] (train.py:32,244)

    def step():
        with torch.no_grad():
            return net.forward_feature(x)

    for _ in range(args.warmup):
        y = step()
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    ev0 = torch.cuda.Event(enable_timing=True)   # torch's current stream == the stream libsrbh launches on
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        y = step()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    assert bool(torch.isfinite(y[0, :, ::16, ::16]).all())
    net.check_status()   # persistent-kernel error word (outside the timed region)
    elapsed, gpu_ms = _max_over_ranks([elapsed, gpu_ms], dev, dist)
    if rank != 0:
        return None
    tiles = B * args.steps * world
    step_s_events = gpu_ms / 1e3 / args.steps
    achieved = GFLOP_PER_TILE_FEATURE * (args.num_block * 5.8886 + 11.19) / 146.630 if args.num_block != 23 \
        else GFLOP_PER_TILE_FEATURE
    tflops = achieved * B / step_s_events / 1e3
    # ---- roofline of the dominant kernel: the persistent trunk (the 345 dense-block convs = 92 % of the FLOPs, ~90 % of
    # the time).  Its launch duration is measured live with HIP events recorded on the stream it is launched on
    # (libsrbh's srbh_trunk_timing hook), over extra forwards outside the timed region.
    import ctypes
    from srbh_amd import _lib
    L = _lib.lib()
    trunk_gflop_tile = args.num_block * 3 * 4096 * 18 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64) / 1e9
    trunk_ms = None
    strict = bool(net._use_strict())          # SRBH_TRUNK_PRECISION=f32 / net.precision: exact-fp32 matrix cores, one launch per conv
    if not strict and L.srbh_trunk_timing(1) == 0:
        acc, nrep = 0.0, max(5, min(20, args.steps))
        try:
            for _ in range(nrep):
                step()
                ms = ctypes.c_float(0.0)
                _lib.check(L.srbh_trunk_last_ms(ctypes.byref(ms)), "srbh_trunk_last_ms")
                acc += ms.value
            trunk_ms = acc / nrep
        except RuntimeError:          # SRBH_PERSISTENT=0 (per-layer launches: no single dominant kernel) -> roofline fields null
            trunk_ms = None
        L.srbh_trunk_timing(0)
    trunk_tflops = trunk_gflop_tile * B / (trunk_ms / 1e3) / 1e3 if trunk_ms else None
    kname = "unknown"
    try:
        kname = L.srbh_trunk_kernel_name().decode()
    except Exception:
        pass
    traffic, traffic_src = (None, None)
    if B == 32 and args.num_block == 23 and not strict and trunk_ms:
        traffic, traffic_src = _recorded_traffic(kname)
    if strict:
        dtype = "f32 operands / f32 accumulate (v_mfma_f32_16x16x4_f32, one launch per conv), f32 residual stream"
        peak = PEAK_F32_MFMA_TFLOPS
    else:
        dtype = "f16 operands / f32 accumulate (MFMA), f32 residual stream"
        peak = PEAK_F16_TFLOPS
    line = {
        "metric": "tiles/sec (64x64x8ch->256x256 height)", "value": round(tiles / elapsed, 2), "unit": "tiles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype,
        "data": "synthetic uniform[0,1) tiles, random-init weights (no datasets/checkpoints offline)",
        "config": {"workload": f"RRDBNet x4 ({args.num_block} RRDB, 64 feat) forward_feature, batch {B} tiles/GPU, "
                               "64x64x3 -> 64x256x256 (BASELINE.json configs[1])",
                   "global_batch": B * world, "parallelism": f"tile-sharded x{world} (no data-path collective)"},
        "roofline": {"bound": "mfma", "achieved": round(trunk_tflops, 2) if trunk_tflops else None, "peak": peak,
                     "unit": "TFLOP/s", "frac": round(trunk_tflops / peak, 4) if trunk_tflops else None,
                     "traffic": traffic,
                     "traffic_source": ((f"{traffic_src} (RECORDED: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                         "command, FETCH doubled per the guide; not measured in this run)") if traffic is not None else traffic_src),
                     "power_capped_peak": None if strict else {
                         "what": "tools/mfma_ceiling.hip on this package (RECORDED at this round's HEAD, profiles/r06d_mfma_ceiling.txt + r06d_mfma_ceiling_smi_summary.txt; package at 1 372 W of 1 400 W under the trunk itself, profiles/r06c_trunk_power_probe.txt): v_mfma_f32_32x32x16_f16 back to back, one wave "
                                 "per SIMD, 256 CUs, real trunk operands, seconds-long runs: MFMA only / + the trunk's LDS read mix / + its weight LDS-DMA stream",
                         "tflops": [1680.7, 1591.6, 1545.0], "frac_of_2500": [0.672, 0.637, 0.618], "sclk_mhz": [1653, 1582, 1586],
                         "frac_of_measured_ceiling": round(trunk_tflops / 1545.0, 4) if trunk_tflops else None},
                     "kernel": ("hconv_f32_kernel per conv (strict fp32: no single dominant kernel; see whole_forward)" if strict else
                                f"{kname} (persistent trunk: 345 dense-block 3x3 convs in one launch)"),
                     "avg_launch_ms": round(trunk_ms, 4) if trunk_ms else None,
                     "algorithmic_gflop_per_launch": round(trunk_gflop_tile * B, 1),
                     "whole_forward": {"achieved": round(tflops, 2), "frac": round(tflops / peak, 4),
                                       "gflop": round(achieved * B, 1), "ms": round(step_s_events * 1e3, 4)}},
    }
    # strict-fp32 GPU path (exact-fp32 matrix cores, one launch per conv) on ONE tile of the sample the CPU leg uses: the on-device
    # yardstick of the fp16-operand fast path; the CPU oracle's output for the same tile comes out of the cpu_baseline leg
    from srbh_amd import synth as _synth
    x1 = _synth.tiles(4, 8, 64, seed=1)[:1, :3].contiguous().to(dev)
    with torch.no_grad():
        y_fast = net.forward_feature(x1).float().contiguous()
        net.precision = "f32"
        y_strict = net.forward_feature(x1).float().contiguous()
        del net.precision
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())      # noqa: E731
    parity = {"output": "forward_feature of 1 tile (64 x 256 x 256)", "tolerance_rel_l2": 1e-3,
              "weights": "random-init (synth 'init' mode: kaiming x 0.1 on the dense-block convs); no trained checkpoint exists offline (SURVEY D8)",
              "vs_strict_f32_gpu_path": {"rel_l2": round(rel(y_fast, y_strict), 7),
                                         "rmse": round(float((y_fast - y_strict).pow(2).mean().sqrt()), 7)},
              "vs_cpu_oracle": None}
    if world == 1 and not args.no_cpu_baseline:
        ref = []
        line["cpu_baseline"] = cpu_baseline(sd, want_ref=ref)
        y_ref = ref[1].to(dev)
        parity["vs_cpu_oracle"] = {"rel_l2": round(rel(y_fast, y_ref), 7),
                                   "rmse": round(float((y_fast - y_ref).pow(2).mean().sqrt()), 7),
                                   "max_abs_over_absmax": round(float((y_fast - y_ref).abs().max() / y_ref.abs().max()), 7),
                                   "strict_f32_gpu_path_rel_l2": round(rel(y_strict, y_ref), 9),
                                   "ref_rms": round(float(y_ref.pow(2).mean().sqrt()), 5)}
    if not args.no_extras and args.num_block == 23 and not strict:
        # the same comparison over several draws (weights x {init, stress} x inputs) against the strict-fp32 GPU path -- itself pinned to
        # ~1e-6 of the CPU oracle (above and tests/test_gpu_rrdbnet.py); the full sweep against the oracle is tests/test_gpu_parity_sweep.py
        sw = []
        xs = torch.cat([_synth.tiles(1, 8, 64, seed=s_)[:, :3] for s_ in (1337, 77)]).contiguous().to(dev)
        for mode in ("init", "stress"):
            for ws in (1337, 1, 2, 3):
                net.load_state_dict(_synth.rrdbnet_state_dict(num_block=args.num_block, seed=ws, mode=mode), strict=True)
                with torch.no_grad():
                    yf = net.forward_feature(xs).float()
                    net.precision = "f32"
                    ys = net.forward_feature(xs).float()
                    del net.precision
                sw += [rel(yf[i:i + 1], ys[i:i + 1]) for i in range(xs.shape[0])]
        net.check_status()
        net.load_state_dict(sd, strict=True)
        sw.sort()
        parity["sweep_max_rel_l2"] = round(sw[-1], 7)
        parity["sweep"] = {"draws": len(sw), "what": "4 weight seeds x {init, stress} x 2 input tiles, fast path vs strict-fp32 GPU path",
                           "min": round(sw[0], 7), "median": round(sw[len(sw) // 2], 7), "max": round(sw[-1], 7)}
    line["parity"] = parity
    return line


def _compact(d, drop=("higher_is_better", "vs_baseline", "warmup", "unit", "scaling", "whole_step")):
    """sub-object of the default line: the standalone `--workload train|predict` lines keep every field and note; here the notes go
    (they are in those lines and in DESIGN.md) and the per-call tables keep 5 rows, so that the whole line stays under the driver's
    8 KB tail"""
    if not d:
        return d
    out = {k: v for k, v in d.items() if k not in drop}
    if isinstance(out.get("parity"), dict):
        out["parity"] = {kk: vv for kk, vv in out["parity"].items() if kk != "what"}
    for k in ("head_roofline", "encdec_kernels", "strict_f32_head", "comm"):
        if isinstance(out.get(k), dict):
            out[k] = {kk: vv for kk, vv in out[k].items() if kk != "note"}
    if isinstance(out.get("head_roofline"), dict) and "kernels" in out["head_roofline"]:
        out["head_roofline"]["kernels"] = [{kk: r[kk] for kk in ("kernel", "calls_per_step", "ms_per_step", "frac_hbm_peak") if kk in r}
                                           for r in out["head_roofline"]["kernels"][:5]]
    if isinstance(out.get("encdec_kernels"), dict):
        so = out["encdec_kernels"].get("stock_ops")
        if isinstance(so, dict):
            out["encdec_kernels"]["stock_ops"] = {"calls": so.get("calls"), "by_site": so.get("by_site")}
    if isinstance(out.get("kernels"), list):
        out["kernels"] = [{kk: r[kk] for kk in ("kernel", "avg_launch_ms", "achieved", "unit", "frac") if kk in r} for r in out["kernels"]]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU per step (configs[1]: 32)")
    ap.add_argument("--num-block", type=int, default=23)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="default run: skip the train_step / predict sub-objects")
    ap.add_argument("--epoch-tiles", type=int, default=31500,
                    help="--workload epoch: tiles of the pass (default 31 500 = 45 000 x 0.7, data/datalist_globe_train_0.7.csv); drop_last")
    ap.add_argument("--details", default=None,
                    help="write the FULL per-call tables (head_roofline / encdec_kernels rows) to this JSON file; the printed line keeps the top rows")
    ap.add_argument("--workload", choices=["feature", "train", "predict", "epoch", "sr_train"], default="feature",
                    help="feature = BASELINE configs[1] (default; carries bounded train_step / predict sub-objects); train = "
                         "configs[2]: full training step, batch 64; epoch = configs[3]: one DP pass over 31 500 synthetic tiles; "
                         "predict = configs[4]: tiled city inference incl. mosaic, one city per step (try --steps 12)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU fallback for the hot path)"
    # test hooks (tests/test_gpu_bench_ranks.py: this file's N>1 code on a ONE-GPU box): SRBH_BENCH_SHARED_DEVICE=1 puts every rank on
    # cuda:0, SRBH_BENCH_BACKEND=gloo carries the collectives without RCCL (which needs one device per rank).  Never set by the driver.
    if os.environ.get("SRBH_BENCH_SHARED_DEVICE", "0") == "1":
        local_rank = 0
    backend = os.environ.get("SRBH_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # self-validation on first contact with a multi-GPU node (round-5 VERDICT item 8): the numbers of an N-GPU line must come from
        # N ranks on N devices over RCCL -- anything else ends the run with a non-zero exit status instead of a plausible line
        info = _dist_info(dist)
        hooked = os.environ.get("SRBH_BENCH_BACKEND") is not None or os.environ.get("SRBH_BENCH_SHARED_DEVICE", "0") == "1"
        problems = []
        if info["world_size_seen"] != args.gpus:
            problems.append(f"the process group has {info['world_size_seen']} ranks, --gpus says {args.gpus}")
        if not hooked:
            if info["backend"] != "nccl":
                problems.append(f"backend is {info['backend']!r}, not 'nccl' (RCCL)")
            if info["rccl_version"] is None:
                problems.append("torch.cuda.nccl.version() is unavailable: no RCCL behind the 'nccl' backend")
            if torch.cuda.device_count() < world:
                problems.append(f"{world} ranks but only {torch.cuda.device_count()} visible devices (one rank per GPU)")
            else:
                # one all-reduce over RCCL right now: every rank contributes its rank + 1
                probe = torch.full((1024,), float(rank + 1), device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if float(probe[0]) != world * (world + 1) / 2 or float(probe[-1]) != world * (world + 1) / 2:
                    problems.append(f"a probe all-reduce returned {float(probe[0])}, expected {world * (world + 1) / 2}")
        if problems:
            sys.exit(f"bench.py --gpus {args.gpus} (rank {rank}): " + "; ".join(problems))

    tb = args.batch if args.batch != 32 else 64
    pb = args.batch if args.batch != 32 else 256      # (256 tiles per batch: +3.7 % over 128 on three interleaved repeats, profiles/r06m_predict_batch_repeats.txt)
    if args.workload == "train":
        line = bench_train(args, rank, world, dev, dist, args.steps, args.warmup, batch=tb, with_kernels=not args.no_extras,
                           with_cpu=not (args.no_extras or args.no_cpu_baseline))
    elif args.workload == "epoch":
        line = bench_train(args, rank, world, dev, dist, 0, args.warmup, batch=tb, epoch_tiles=args.epoch_tiles, with_kernels=False,
                           with_cpu=False)
    elif args.workload == "predict":
        line = bench_predict(args, rank, world, dev, dist, args.steps, args.warmup, batch=pb)
    elif args.workload == "sr_train":
        line = bench_sr_train(args, rank, world, dev, dist, args.steps, args.warmup, batch=args.batch if args.batch != 32 else 8)
    else:
        line = bench_feature(args, rank, world, dev, dist)
        if not args.no_extras and args.num_block == 23 and args.batch == 32:
            extras = {}
            for key, fn in (("train_step", lambda: bench_train(args, rank, world, dev, dist, 20, 8, batch=64,      # (8 warm-up + 20 timed steps: the pipelined step needs a few steps to settle; 5 + 10 read 1.8 ms above the 40-step figure)
                                                                with_cpu=not args.no_cpu_baseline)),
                            ("predict", lambda: bench_predict(args, rank, world, dev, dist, int(os.environ.get("SRBH_BENCH_PREDICT_CITIES", "301")), 1, batch=256))):   # configs[4] at its stated size: all 301 cities (~225 s on one GPU)
                try:
                    extras[key] = _compact(fn())
                except Exception as e:          # the headline must survive a failing extra (and say so)
                    extras[key] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            if line is not None:
                line.update(extras)
                # the driver keeps the TAIL of this line: the figures a reader needs from the sub-objects, once more, last
                t, p_ = extras.get("train_step") or {}, extras.get("predict") or {}
                par = (line.get("parity") or {}).get("vs_cpu_oracle") or {}
                line["summary"] = {
                    "feature_b32": {"tiles_per_s": line["value"], "ms_per_step": line["ms_per_step"], "trunk_frac_mfma_peak": line["roofline"]["frac"],
                                    "trunk_ms": line["roofline"]["avg_launch_ms"], "rel_l2_vs_cpu_oracle": par.get("rel_l2"),
                                    "sweep_max_rel_l2": (line.get("parity") or {}).get("sweep_max_rel_l2"),
                                    "weights": "random-init (no trained checkpoint offline)"},
                    "train_step_b64": {"error": t["error"]} if "error" in t else {
                        "credited_strict_f32_head": {"ms_per_step": (t.get("strict_f32_head") or {}).get("ms_per_step"),
                                                     "tiles_per_s": (t.get("strict_f32_head") or {}).get("value")},
                        "fast_mode_f16_head": {"ms_per_step": t.get("ms_per_step"), "tiles_per_s": t.get("value"),
                                               "gradient_parity_vs_exact": {k: (t.get("parity") or {}).get(k) for k in
                                                                            ("whole_gradient", "heads_max_rel_l2", "within_tolerance", "tolerance", "error")
                                                                            if (t.get("parity") or {}).get(k) is not None}},
                        "head_frac_hbm_peak": (t.get("head_roofline") or {}).get("frac_hbm_peak"),
                        "head_ms": (t.get("head_roofline") or {}).get("ms_per_step"),
                        "encdec_libsrbh_ms": (t.get("encdec_kernels") or {}).get("ms_per_step"),
                        "encdec_stock_op_calls_2steps": ((t.get("encdec_kernels") or {}).get("stock_ops") or {}).get("calls"),
                        "cpu_baseline_tiles_per_s": (t.get("cpu_baseline") or {}).get("value")},
                    "predict_cities": {"error": p_["error"]} if "error" in p_ else {
                        "cities": (p_.get("config") or {}).get("cities"), "of": 301, "tiles": (p_.get("config") or {}).get("tiles"),
                        "tiles_per_s": p_.get("value"), "p50_city_latency_ms": p_.get("p50_city_latency_ms"),
                        "p95_city_latency_ms": p_.get("p95_city_latency_ms"), "max_city_latency_ms": p_.get("max_city_latency_ms"),
                        "frac_mfma_peak_per_gpu": (p_.get("roofline") or {}).get("frac")},
                    # the fwd+bwd curve of BASELINE's metric at THIS N (configs[2] at N=1, configs[3]'s step at N>1) and what the process
                    # group was: a SCALE record carries the gradient all-reduce whatever the headline workload is
                    "dp_train": {"error": t["error"]} if "error" in t else dict(
                        {"tiles_per_s": t.get("value"), "ms_per_step": t.get("ms_per_step"), "n_gpus": world,
                         "comm_ms": (t.get("comm") or {}).get("comm_ms"), "exposed_comm_ms": (t.get("comm") or {}).get("exposed_comm_ms"),
                         "buckets": (t.get("comm") or {}).get("buckets"), "grad_bytes": (t.get("comm") or {}).get("grad_bytes"),
                         "prefetch_hits_per_rank": (t.get("comm") or {}).get("prefetch_hits_per_rank")},
                        **_dist_info(dist)),
                }
    if rank == 0:
        if args.details and _DETAILS:
            with open(args.details, "w") as f:
                json.dump(_DETAILS, f, indent=1)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
