"""developer aid (round-3 VERDICT task 7a): what ONE rank's host thread spends issuing a data-parallel training step when it owns
exactly one core -- the question for an 8-rank node, where every rank queues ~1 300 launches per step.

  leg 1: one rank, pinned to one core, the real step (B=64, 23 blocks, persistent trunk), eager launches vs the replayed graph;
  leg 2: two ranks sharing this box's one GPU (gloo; per-layer trunk, as two processes cannot both own every CU), each pinned to
         its own core: eager step with hook-launched buckets vs TrainStep(graph=True) (replay, then buckets + Adam).

"issue" = perf_counter until ts() returns with no synchronisation inside the loop (the queue absorbs the launches), "cpu" =
process_time per step, "wall" = including the final synchronize.  gloo's all-reduce stages through the host, so leg 2's `finish`
share is NOT what RCCL costs (one enqueue per bucket); it is listed apart.

usage (GPU box): python tools/host_issue_dp.py [steps]"""
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def _nets(dev, blocks):
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    torch.manual_seed(0)
    net_hr = RRDBNet(3, 3, num_block=blocks).to(dev)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7).to(dev)
    return net_hr, net


def _time(ts, batch, steps, warm):
    for _ in range(warm):
        ts(batch)
    torch.cuda.synchronize()
    fin = 0.0
    red = ts.reducer
    if red is not None:                       # time finish() apart (gloo stages through the host)
        inner = red.finish

        def timed():
            nonlocal fin
            t = time.perf_counter()
            r = inner()
            fin += time.perf_counter() - t
            return r
        red.finish = timed
    c0, t0 = time.process_time(), time.perf_counter()
    for _ in range(steps):
        ts(batch)
    t_issue, c_issue = time.perf_counter() - t0, time.process_time() - c0
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    if red is not None:
        red.finish = inner
    k = 1e3 / steps
    return {"issue_ms": round(t_issue * k, 2), "cpu_ms": round(c_issue * k, 2), "wall_ms": round(t_wall * k, 2),
            "finish_ms": round(fin * k, 2), "issue_minus_finish_ms": round((t_issue - fin) * k, 2)}


def _rank(rank, world, port, steps, core, q):
    os.sched_setaffinity(0, {core})
    torch.set_num_threads(1)
    from srbh_amd.harness import TrainStep, synthetic_batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    net_hr, net = _nets(dev, 23)
    batch = synthetic_batch(64, 1337 + rank, dev)
    out = {}
    for graph in (False, True):
        ts = TrainStep(net_hr, net, dev, world=world, status_every=0, graph=graph)
        out["graph" if graph else "eager"] = _time(ts, batch, steps, 6 if graph else 3)
        if ts.reducer is not None:
            ts.reducer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    q.put((rank, core, out))


def main():
    import json
    import torch.multiprocessing as mp
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    cores = sorted(os.sched_getaffinity(0))
    ctx = mp.get_context("spawn")
    res = {"host_cores_visible": len(cores)}
    for world in (1, 2):
        os.environ["SRBH_PERSISTENT"] = "1" if world == 1 else "0"
        q = ctx.Queue()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=_rank, args=(r, world, port, steps, cores[(1 + r) % len(cores)], q)) for r in range(world)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=1500) for _ in procs)
        for p in procs:
            p.join(60)
        res[f"world{world}"] = [{"rank": r, "core": c, **o} for r, c, o in got]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
