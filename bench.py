#!/usr/bin/env python3
"""bench.py -- tiles/s of the MI355X-native hot path (BASELINE.json metric) + roofline + CPU baseline.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (default, BASELINE.json configs[1]): RRDBNet x4 (23 RRDB, 64 feat) forward_feature, batch 32
synthetic 64x64 tiles per GPU, inputs resident in HBM, random-init weights of the reference architecture.
A "step" is one forward_feature over one batch.  Tiles are independent, so ranks shard the tile stream with
no data-path collective ("scaling": "weak"); the only collectives are the timing barrier / max.

One JSON line is printed by rank 0 (contract in the task statement); extra keys:
  roofline     -- MFMA roofline of the conv stack: 146.630 GFLOP per tile (hook-counted on the
                  reference) x tiles per step / step duration measured with HIP events on the launch stream.
  cpu_baseline -- the CPU oracle (oracle/srbh_oracle.py, proven equal to the reference) timed on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

GFLOP_PER_TILE_FEATURE = 146.630   # 2*9*Cin*Cout*H*W over the 350 convs of forward_feature
PMC_JSON = "r01f_pmc_hbm_traffic.json"   # written by tools/pmc_traffic.py from the rocprofv3 --pmc passes
PEAK_F16_TFLOPS = 2500.0           # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md), never the sparse figure


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


def cpu_baseline(sd, seconds_budget=20.0):
    """Oracle forward_feature on the host cores, bounded to ~seconds_budget of wall time.
    oneDNN scales badly past a few dozen threads on these small convs, so the thread count is capped at 32
    (`cores` reports the threads actually used).  Protocol: one timed B=1 call decides the sample size."""
    from oracle import srbh_oracle as O, synth
    cores = min(_host_cores(), 32)
    torch.set_num_threads(cores)
    x = synth.tiles(4, 8, 64, seed=1)[:, :3].contiguous()
    sd_cpu = {k: v.float() for k, v in sd.items()}
    t0 = time.perf_counter()
    O.rrdbnet_forward_feature(sd_cpu, x[:1])  # first call also creates the oneDNN primitives
    t_first = time.perf_counter() - t0
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


def bench_train(args, rank, world, dev, dist):
    """BASELINE configs[2]/[3]: RRDBNet forward (no grad) + SRRegress_Cls_feature forward/backward + Adam, batch 64 per
    GPU, gradients averaged over ranks by one bucketed RCCL all-reduce per step."""
    from oracle import synth
    from srbh_amd.harness import TrainStep, synthetic_batch
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    B = args.batch if args.batch != 32 else 64
    net_hr = RRDBNet(3, 3, num_block=args.num_block)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=args.num_block, seed=1337, mode="init"))
    torch.manual_seed(1337)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True,
                                chans_build=7)
    sync_bn = os.environ.get("SRBH_SYNC_BN", "0") == "1"        # default: per-rank BatchNorm statistics
    ts = TrainStep(net_hr.to(dev), net.to(dev), dev, world=world, sync_bn=sync_bn)
    batch = synthetic_batch(B, 1337 + rank, dev)
    for _ in range(args.warmup):
        ts(batch)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = ts(batch)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    if rank == 0:
        gf_tile = 146.63 + 8.12 * 3 + 0.9 * 3      # RRDB fwd + head fwd/bwd (~3x fwd) + encoder/decoders
        tf = gf_tile * B * world * args.steps / elapsed / 1e3
        print(json.dumps({
            "metric": "tiles/sec (64x64x8ch->256x256 height) fwd+bwd", "value": round(B * world * args.steps / elapsed, 2),
            "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 operands/f32 acc (RRDB), f32 (head fwd+bwd)", "data": "synthetic",
            "config": {"workload": f"full train step: RRDBNet fwd (no grad) + SRRegress_Cls_feature fwd/bwd + Adam, batch {B}/GPU "
                                   "(BASELINE.json configs[2])", "global_batch": B * world,
                       "parallelism": f"dp{world} (bucketed RCCL grad all-reduce)"},
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / PEAK_F16_TFLOPS, 4), "traffic": None,
                         "kernel": "whole step (RRDB f16 MFMA stack dominates the FLOPs)"},
            "final_loss": float(loss)}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_predict(args, rank, world, dev, dist):
    """BASELINE configs[4]: urban-centre tiled inference.  A "step" is ONE synthetic city: its grid cells (64x64x8 LR tiles,
    stride 48 as in the reference's grid loader) are sharded over the ranks, each rank runs RRDBNet features -> eval head ->
    quantise + integer mosaic on the device, the integer mosaics are summed over ranks (bit-identical to the serial
    result) and finalised (argmax / normalise).  City sizes: the first --steps (+ warm-up) of 301 cell counts drawn
    log-uniform in [200, 20000] (numpy default_rng(2024)); no per-city grid counts ship with the reference.
    GeoTIFF IO is outside the path."""
    import numpy as np
    from oracle import synth
    from srbh_amd.harness import predict_tiles
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.mosaic import Mosaic
    from srbh_amd.rrdbnet import RRDBNet
    net_hr = RRDBNet(3, 3, num_block=args.num_block)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=args.num_block, seed=1337, mode="init"))
    net_hr = net_hr.to(dev).eval()
    torch.manual_seed(1337)
    model = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=False,
                                  chans_build=7).to(dev).eval()
    counts = np.exp(np.random.default_rng(2024).uniform(np.log(200.0), np.log(20000.0), 301)).astype(int)
    # MIOpen solver search for the stock-op encoder / decoders: without it a few forward convolutions run on MIOpen's naive
    # kernels (immediate-mode fallback).  One-time cost in the warm-up cities; every batch has the same shape.  (Not used
    # for the training workload: the backward searches take minutes.)
    torch.backends.cudnn.benchmark = os.environ.get("SRBH_MIOPEN_FIND", "1") == "1"
    batch = args.batch if args.batch != 32 else 128     # 288 GB of HBM: larger batches amortise the stock-op encoder's small launches
    n_warm = min(args.warmup, 2)
    todo = [int(c) for c in counts[:args.steps]]
    pad_to = 32 if batch % 32 == 0 and batch > 32 else None          # ragged tails run as 32/64/96/... tiles, not the full batch
    warm = [min(int(c), 256) for c in counts[-n_warm:]] if n_warm else []
    if pad_to and args.warmup:
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
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        del m, out, tiles
        return dt

    for i, n in enumerate(warm):
        city(n, 99 + i)
    lat = [city(n, 2024 + i) for i, n in enumerate(todo)]
    if rank == 0:
        total, elapsed = sum(todo), sum(lat)
        order = sorted(range(len(lat)), key=lambda i: lat[i])
        mid = order[len(order) // 2]
        print(json.dumps({
            "metric": "tiles/sec (64x64x8ch->256x256 height) tiled inference incl. quantise + mosaic", "value": round(total / elapsed, 2),
            "unit": "tiles/s", "n_gpus": world, "steps": len(todo), "warmup": args.warmup, "ms_per_step": round(elapsed / len(todo) * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16 operands/f32 acc (RRDB), f32 (head), integer mosaic", "data": "synthetic cities, random-init weights",
            "config": {"workload": f"sliding-window predict path, {len(todo)} of 301 synthetic cities (cells log-uniform 200..20000, "
                                   f"seed 2024), batch {batch}/GPU (BASELINE.json configs[4])",
                       "cities": len(todo), "tiles": total, "parallelism": f"each city's cells sharded x{world}, integer mosaic row bands gathered on rank 0"},
            "p50_city_latency_ms": round(lat[mid] * 1e3, 2), "p50_city_tiles": todo[mid],
            "max_city_latency_ms": round(max(lat) * 1e3, 2), "max_city_tiles": max(todo)}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU per step (configs[1]: 32)")
    ap.add_argument("--num-block", type=int, default=23)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["feature", "train", "predict"], default="feature",
                    help="feature = BASELINE configs[1] (default); train = configs[2]: full training step, batch 64; "
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from oracle import synth  # synthetic weights/inputs only (the checker itself runs in cpu_baseline)
    from srbh_amd.rrdbnet import RRDBNet

    if args.workload == "train":
        return bench_train(args, rank, world, dev, dist)
    if args.workload == "predict":
        return bench_predict(args, rank, world, dev, dist)

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

    if dist is not None:
        t = torch.tensor([elapsed, gpu_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, gpu_ms = float(t[0]), float(t[1])

    if rank == 0:
        tiles = B * args.steps * world
        step_s_events = gpu_ms / 1e3 / args.steps
        achieved = GFLOP_PER_TILE_FEATURE * (args.num_block * 5.8886 + 11.19) / 146.630 if args.num_block != 23 \
            else GFLOP_PER_TILE_FEATURE
        tflops = achieved * B / step_s_events / 1e3
        # ---- roofline of the dominant kernel: ptrunk_kernel (the 345 dense-block convs = 92 % of the FLOPs, ~90 % of the
        # time).  Its launch duration is measured live with HIP events recorded on the stream it is launched on
        # (libsrbh's srbh_trunk_timing hook), over extra forwards outside the timed region.
        from importlib import import_module
        _lib = import_module("srbh_amd._lib")
        L = _lib.lib()
        trunk_gflop_tile = args.num_block * 3 * 4096 * 18 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64) / 1e9
        trunk_ms = None
        if L.srbh_trunk_timing(1) == 0:
            import ctypes
            acc, nrep = 0.0, max(5, min(20, args.steps))
            for _ in range(nrep):
                step()
                ms = ctypes.c_float(0.0)
                _lib.check(L.srbh_trunk_last_ms(ctypes.byref(ms)), "srbh_trunk_last_ms")
                acc += ms.value
            L.srbh_trunk_timing(0)
            trunk_ms = acc / nrep
        trunk_tflops = trunk_gflop_tile * B / (trunk_ms / 1e3) / 1e3 if trunk_ms else None
        traffic = None   # HBM bytes per ptrunk launch from the PMC passes recorded under profiles/ (same command, B=32)
        try:
            if B == 32 and args.num_block == 23:
                traffic = json.load(open(os.path.join(ROOT, "profiles", PMC_JSON)))["dominant_kernel_hbm_bytes_per_launch"]
        except Exception:
            pass
        line = {
            "metric": "tiles/sec (64x64x8ch->256x256 height)", "value": round(tiles / elapsed, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 operands / f32 accumulate (MFMA), f32 residual stream",
            "data": "synthetic uniform[0,1) tiles, random-init weights (no datasets/checkpoints offline)",
            "config": {"workload": f"RRDBNet x4 ({args.num_block} RRDB, 64 feat) forward_feature, batch {B} tiles/GPU, "
                                   "64x64x3 -> 64x256x256 (BASELINE.json configs[1])",
                       "global_batch": B * world, "parallelism": f"tile-sharded x{world} (no data-path collective)"},
            "roofline": {"bound": "mfma", "achieved": round(trunk_tflops, 2) if trunk_tflops else None, "peak": PEAK_F16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(trunk_tflops / PEAK_F16_TFLOPS, 4) if trunk_tflops else None,
                         "traffic": traffic,
                         "kernel": "ptrunk_kernel (persistent trunk: 345 dense-block 3x3 convs in one launch)",
                         "avg_launch_ms": round(trunk_ms, 4) if trunk_ms else None,
                         "algorithmic_gflop_per_launch": round(trunk_gflop_tile * B, 1),
                         "whole_forward": {"achieved": round(tflops, 2), "frac": round(tflops / PEAK_F16_TFLOPS, 4),
                                           "gflop": round(achieved * B, 1), "ms": round(step_s_events * 1e3, 4)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
