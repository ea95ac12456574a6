"""Callers of the hot path, reproduced just far enough to drive and measure it.

The reference's train.py / predict script are out of scope; this module restates their *call sequence*
around the two networks so that a full training step (BASELINE config 3), its data-parallel form (config 4: one
process per GPU, gradients averaged by one bucketed RCCL all-reduce per step) and tile-sharded inference (config 5) can
be exercised.  Losses follow losses_pytorch/selfloss.py:6-17,81-91,145-168 in plain torch ops (they are the "next"
row 8f-3, not yet kernels).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, wcache

HIR = (0, 3, 12, 21, 30, 60, 90, 256)                       # train.py:55
# hierweight(bh_stats_globe, HIR) as probed on the reference data; synthetic labels reuse it
CLASS_WEIGHT = (0.08878965, 0.272375, 0.32563883, 0.74654097, 0.99655239, 1.62749907, 2.94260409)


from .losses import MSE_adapt_weight, CE_DICE_adapt_weight   # noqa: E402  (libsrbh reductions, losses.py)


def synthetic_batch(batch, seed, device, aggregate=None):
    """One training batch shaped like myImageFloder_S12_globe's output (BH_loader.py:338-400): 8-channel 64x64 tile in
    [0,1), 256x256 height label with ~82% zeros in blocky 8x8 patches, hierarchy class map, per-pixel class weights and
    their 4x4 aggregates.  ``aggregate`` defaults to the libsrbh aggregate_torch kernel on `device`."""
    g = torch.Generator()
    g.manual_seed(seed)
    lr = torch.rand(batch, 8, 64, 64, generator=g)
    coarse = torch.rand(batch, 1, 32, 32, generator=g)
    hval = torch.rand(batch, 1, 32, 32, generator=g) ** 3 * 120.0
    height = torch.where(coarse > 0.82, hval, torch.zeros_like(hval))
    height = F.interpolate(height, scale_factor=8, mode="nearest").round()
    edges = torch.tensor(HIR[1:-1], dtype=torch.float32)
    build = torch.bucketize(height[:, 0], edges, right=True)
    build = torch.where(height[:, 0] <= 0, torch.zeros_like(build), build).long().clamp_(0, 6)
    weight = torch.tensor(CLASS_WEIGHT)[build]
    lr, height, build, weight = lr.to(device), height.to(device), build.to(device), weight.to(device)
    if aggregate is None:
        from .aggregate import aggregate_torch as aggregate
    height_aggre = aggregate(height, 0.25).reshape(batch, 64, 64)
    weight_aggre = aggregate(weight[:, None].contiguous(), 0.25).reshape(batch, 64, 64)
    return lr, height[:, 0], height_aggre, build, weight, weight_aggre


def shard_range(n_items, rank, world):
    """contiguous, balanced shard of range(n_items) for `rank` (tiles are independent: no data-path collective)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_grads(params, world, dist=None, bucket_bytes=64 << 20):
    """Average gradients over ranks with few large all-reduces AFTER backward (RCCL over xGMI is per-link bound: prefer big
    buckets).  Parameters without a gradient (e.g. the unused encoder._conv_head) are skipped on every rank alike.
    This is the simple post-backward sweep; `GradReducer` below overlaps the same reduction with backward."""
    if world <= 1:
        return 0
    if dist is None:
        import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    buckets, cur, size = [], [], 0
    for gr in grads:
        cur.append(gr)
        size += gr.numel() * gr.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch.cat([gr.reshape(-1) for gr in b])
        dist.all_reduce(flat)
        flat.div_(world)
        off = 0
        for gr in b:
            n = gr.numel()
            gr.copy_(flat[off:off + n].view_as(gr))
            off += n
    return len(buckets)


class GradReducer:
    """Gradient all-reduce overlapped with backward (one process per GPU, RCCL over xGMI).

    Step 1 runs the post-backward sweep and RECORDS the order in which gradients became ready (autograd
    post-accumulate hooks); from it a static bucket plan is built -- parameters grouped in ready order into buckets of
    ~`bucket_bytes`, parameters that never receive a gradient (encoder._conv_head / _bn1) left out on every
    rank alike.  From step 2 on, the hook of a bucket's LAST parameter copies that bucket's gradients into its
    persistent flat buffer and launches ONE asynchronous all-reduce, so the early buckets (heads, decoders: the first
    gradients backward produces) travel while the encoder's backward still runs; `finish()` waits, divides by the
    world size and scatters the averages back into `.grad`.  xGMI is point-to-point (ring collectives are per-link
    bound), so buckets are tens of MB: large enough to run at link rate, small enough that only the last one is
    exposed.  `exposed_ms` (timing=True) is the device time between the end of backward and the last bucket landing."""

    def __init__(self, params, world, dist=None, bucket_bytes=24 << 20, timing=False):
        self.params = [p for p in params if p.requires_grad]
        self.world = world
        if dist is None and world > 1:
            import torch.distributed as dist
        self.dist = dist
        self.bucket_bytes = bucket_bytes
        self.timing = timing
        self.plan = None             # list of buckets: [(param, offset, numel)], flat buffer, trigger param id
        self._order = []
        self._ready = {}
        self._work = []
        self._hooks = []
        self.exposed_ms = []
        self.n_buckets = 0
        self.paused = False          # True while TrainStep records its graph: the hooks fire during the capture and must not launch
        if world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    # ---- autograd hook: runs as soon as p.grad is final for this backward
    def _on_grad(self, p):
        if self.paused:
            return
        if self.plan is None:
            self._order.append(p)
            return
        b = self._bucket_of.get(id(p))
        if b is None:
            return
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _launch(self, b):
        flat = b["flat"]
        torch._foreach_copy_([flat[o:o + n].view_as(q.grad) for q, o, n in b["items"]], [q.grad for q, _, _ in b["items"]])
        self._work.append((b, self.dist.all_reduce(flat, async_op=True)))

    def _agree_on_order(self):
        """The bucket layout comes from THIS rank's autograd ready order; flat buffers of different ranks must line up the same
        parameters.  Like DDP, rank 0's order wins: its sequence of parameter indices is broadcast and every rank re-orders to
        it; a rank whose SET of gradient-receiving parameters differs raises instead of averaging unrelated gradients."""
        index = {id(p): i for i, p in enumerate(self.params)}
        mine = [index[id(p)] for p in self._order]
        # (the device of the PARAMETERS, also on a rank that received no gradient: a CPU control tensor under RCCL raises or hangs
        # before the intended "ranks disagree" error -- round-3 ADVICE)
        dev = self.params[0].device if self.params else torch.device("cpu")
        n = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
        self.dist.broadcast(n, src=0)
        ref = torch.tensor(mine if len(mine) == int(n) else [0] * int(n), dtype=torch.int64, device=dev)
        self.dist.broadcast(ref, src=0)
        ref = [int(v) for v in ref.tolist()]
        ok = torch.tensor([1 if sorted(ref) == sorted(mine) else 0], dtype=torch.int64, device=dev)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
        if int(ok) == 0:
            raise RuntimeError("GradReducer: the ranks disagree on WHICH parameters receive gradients (rank 0 has %d, this rank %d); "
                               "the bucketed all-reduce would average unrelated tensors" % (len(ref), len(mine)))
        self._order = [self.params[i] for i in ref]

    def _build_plan(self):
        buckets, cur, size = [], [], 0
        for p in self._order:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
        if cur:
            buckets.append(cur)
        self.plan, self._bucket_of = [], {}
        for ps in buckets:
            items, off = [], 0
            for q in ps:
                items.append((q, off, q.numel()))
                off += q.numel()
            b = {"items": items, "flat": torch.empty(off, dtype=ps[0].dtype, device=ps[0].device), "pending": len(ps), "n": len(ps)}
            self.plan.append(b)
            for q in ps:
                self._bucket_of[id(q)] = b
        self.n_buckets = len(self.plan)

    def finish(self):
        """Call after backward, before optimizer.step()."""
        if self.world <= 1:
            return 0
        if self.plan is None:        # first step: plain sweep in ready order, then freeze the bucket plan
            seen = set()
            self._order = [p for p in self._order if p.grad is not None and not (id(p) in seen or seen.add(id(p)))]
            self._agree_on_order()       # (before the first sweep: its buckets are cut from this order too)
            allreduce_grads(self._order, self.world, self.dist, self.bucket_bytes)
            self._build_plan()
            return self.n_buckets
        ev0 = ev1 = None
        if self.timing and self.plan[0]["flat"].is_cuda:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for b in self.plan:          # a bucket whose parameters did not all fire (should not happen: static graph)
            if b["pending"] != 0:
                if b["pending"] != b["n"] or any(q.grad is None for q, _, _ in b["items"]):
                    raise RuntimeError("GradReducer: the set of parameters receiving gradients changed between steps")
                self._launch(b)
        for b, w in self._work:
            w.wait()                 # (device tensors: the current stream waits for the collective; gloo: host wait)
            b["flat"].div_(self.world)
            torch._foreach_copy_([q.grad for q, _, _ in b["items"]], [b["flat"][o:o + n].view_as(q.grad) for q, o, n in b["items"]])
            b["pending"] = b["n"]
        self._work = []
        if ev0 is not None:
            ev1.record()
            self._last_events = (ev0, ev1)
        return self.n_buckets

    def pop_exposed_ms(self):
        """device time finish() spent behind backward in the last step (timing=True; synchronises)."""
        ev = getattr(self, "_last_events", None)
        if ev is None:
            return None
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1])

    def isolated_comm_ms(self, reps=3):
        """the all-reduces of one step on their own (nothing to overlap with): what the overlap has to hide."""
        if self.world <= 1 or not self.plan:
            return 0.0
        import time
        cuda = self.plan[0]["flat"].is_cuda
        best = None
        for _ in range(reps):
            if cuda:
                torch.cuda.synchronize()
            self.dist.barrier()
            t0 = time.perf_counter()
            for b in self.plan:
                self.dist.all_reduce(b["flat"])
            if cuda:
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        return best


# RRDBNet features handed to the head as an fp16 channels_last tensor whenever the head's convolutions run their fp16-operand forms
# (inference chain, TrainStep's 'f16' mode): conv_hr rounds once in its epilogue -- exactly what the head's entry kernel did while
# staging the fp32 tensor, so the forward values do not change by a bit -- and conv_hr's write, the entry's read and the two entry
# weight gradients' reads halve (round-3 VERDICT, What's weak #7).  SRBH_FEATURE_H16=0: the fp32 hand-off (A/B aid).
FEATURE_H16 = os.environ.get("SRBH_FEATURE_H16", "1") == "1"


def features_for_head(net_hr, x, h16=None, model=None):
    """net_hr.forward_feature(x) in the element type the head will stage: fp16 when its fp16-operand kernels are active (h16: the
    caller's hrfuse.head_h16() in the grad mode the HEAD will run in; None = right now), the trunk is on its fast path AND the
    consumer is this package's head (`model.hrfeat` an hrfuse.HRfeature: any other module -- the reference / oracle nn.Module, a user
    head -- keeps getting the fp32 tensor of the nn.Module API); SRBH_PTAIL=0 (no fp16 store outside the persistent tail kernel)
    also falls back to fp32 instead of raising."""
    from . import hrfuse as _H
    if h16 is None:
        h16 = _H.head_h16()
    consumer_ok = model is None or isinstance(getattr(model, "hrfeat", None), _H.HRfeature)
    if (FEATURE_H16 and h16 and consumer_ok and os.environ.get("SRBH_PTAIL", "1") != "0"
            and hasattr(net_hr, "_use_strict") and not net_hr._use_strict()
            and not (getattr(net_hr, "_train_path", False) and torch.is_grad_enabled())):
        return net_hr.forward_feature(x, out_dtype=torch.float16)
    return net_hr.forward_feature(x)


class _Prefetch:
    """RRDBNet features of a batch, being computed on the trunk stream: calling it makes the current stream wait for them"""

    def __init__(self, lr, fea, done):
        # the announced tensor ITSELF is held: its storage cannot be freed and handed out again at the same address with version 0
        # (a pointer + version + shape key would then match a DIFFERENT batch and serve it the wrong features, round-5 ADVICE)
        self.lr, self.version = lr, lr._version
        self.fea, self.done = fea, done

    def matches(self, lr):
        return lr is self.lr and lr._version == self.version

    def abandon(self):
        """the features will not be used: the current stream still has to wait for the launches that compute them, because an inline
        forward_feature that follows shares the RRDBNet's workspace and progress flags (keyed by geometry, not by stream)"""
        torch.cuda.current_stream(self.fea.device).wait_event(self.done)

    def __call__(self):
        cur = torch.cuda.current_stream(self.fea.device)
        cur.wait_event(self.done)
        self.fea.record_stream(cur)
        return self.fea


def pipe_images(device):
    """images per trunk launch of a pipelined step: the persistent trunk kernel takes one CU per workgroup (8 per image) with all of its
    LDS, so a launch on HALF of the CUs leaves the other half to the kernels it runs beside (SRBH_PIPE_IMAGES overrides).  Measured on
    256 CUs at batch 64 (profiles/r05ac_ab_pipeline.txt; serial step 30.8 ms): 4 launches of 16 images 27.5 ms, 3 launches of 21-22
    29.5 ms, launches of 32 (the whole chip: nothing else gets in) no gain, launches of <= 12 slower than the serial step."""
    e = os.environ.get("SRBH_PIPE_IMAGES")
    if e:
        return max(1, int(e))
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    return max(1, ncu // 16)


class TrainStep:
    """train.py:133-179,243-256: frozen RRDBNet feature extractor + trainable SRRegress_Cls_feature, three
    uncertainty-weighted losses, Adam(lr 1e-3, wd 1e-4) with the log_vars as an extra param group.
    world > 1: gradients are averaged by `GradReducer` (bucketed all-reduce launched from autograd hooks while backward
    is still running); ``overlap=False`` selects the plain post-backward sweep.

    ``step(batch, next_batch=...)`` (round 5, the pipelined step): the frozen RRDBNet's features of the NEXT batch are computed on a
    second stream while this step's backward walks the encoder / decoders and the next step's forward walks them again -- two
    chains of ~10-30 us kernels that leave most of the chip idle -- in launches of `pipe_images()` images, so that the trunk's
    workgroups (one CU each) leave CUs to those chains.  The trunk goes out when backward has left the head's chip-filling kernels
    (a post-accumulate hook on HRfeature's first weight).  Same arithmetic per batch; features are those of the batch they are
    used for (matched by tensor identity + version)."""

    def __init__(self, net_hr, net, device, world=1, lr=1e-3, sync_bn=False, overlap=True, timing=False, status_every=100,
                 head_precision="f16", graph=False):
        # mixed precision of the head's convolutions while training (hrfuse.set_head_precision): "f16" = forward convs with
        # fp16 operands, data and weight gradients with bf16 operands (fp32's exponent range: no loss scaling), fp32 accumulation
        # everywhere, BatchNorm, losses and Adam in fp32 -- what the north star's "fp16 MFMA, <= 1e-3 on the height maps" buys;
        # "f32" = the exact-fp32 head the parity tests pin; "auto" = leave the module default (exact while a graph is recorded)
        # Scoped to the step (hrfuse.head_precision context): constructing a TrainStep leaves the process-wide mode alone.
        from . import hrfuse as _H
        if head_precision not in ("auto", "f16", "f32"):
            raise ValueError("head precision must be 'auto', 'f16' or 'f32'")
        if graph and world > 1 and sync_bn:
            raise ValueError("TrainStep(graph=True, sync_bn=True, world > 1): the BatchNorm all-reduces inside the forward are not captured; "
                             "use graph=False")
        self._H = _H
        self.head_precision = head_precision
        if sync_bn and world > 1:
            # global-batch BatchNorm statistics (what the reference computes on one device): libsrbh BatchNorms all-reduce
            # their partial sums (hrfuse.set_bn_sync), the stock-op encoder / decoders become torch SyncBatchNorm
            from . import hrfuse
            hrfuse.set_bn_sync(world)
            for name in ("encoder", "decoder1", "decoder2"):
                if hasattr(net, name):
                    setattr(net, name, nn.SyncBatchNorm.convert_sync_batchnorm(getattr(net, name)))
        self.net_hr, self.net, self.world = net_hr.eval(), net.train(), world
        for p in self.net_hr.parameters():
            p.requires_grad_(False)
        self.criterion = [MSE_adapt_weight(device=device), MSE_adapt_weight(device=device),
                          CE_DICE_adapt_weight(device=device)]
        # graph=True (one GPU; optional, NOT faster than eager launches since the optimizer is fused: 49.0 vs 49.1 ms): after three
        # eager steps the WHOLE step -- RRDBNet features, model forward, losses, backward, Adam --
        # is captured into one HIP graph and replayed (Adam with capturable=True keeps its step counters on the device).  libsrbh's
        # own calls are replay-safe (device state is cleared by kernels: hipMemsetAsync nodes of a replayed graph are not ordered
        # behind the previous replay's kernels on ROCm 7.2); the stock ops of the step still contain a few device-to-device memcpy
        # nodes (ATen clone / copy_), for which the same caution applies -- synchronise between replays if in doubt.
        self.use_graph = bool(graph)
        # fused=True on the GPU: the whole Adam update is a handful of multi-tensor launches (the default foreach path is ~50
        # launches and 5.5 ms of host time per step; with capturable=True its bias-correction pow even falls back to one launch
        # per parameter, +800 launches in the captured graph).  Same update rule (torch/optim/adam.py), fp32 state.
        fused = torch.device(device).type == "cuda"
        # round 5: the same update as ONE libsrbh launch over a device table of tensors (srbh_amd.optim.Adam, csrc/srbh_optim.hip) -- eager steps
        # on a ROCm device; a captured step keeps torch's capturable optimizer (its step counter lives on the device).  SRBH_ADAM=0: torch's.
        if fused and not self.use_graph and os.environ.get("SRBH_ADAM", "1") == "1":
            from .optim import Adam as _Adam
            self.optimizer = _Adam(net.parameters(), lr=lr, weight_decay=1e-4)
        else:
            self.optimizer = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=1e-4, capturable=self.use_graph and world == 1, fused=fused)
        self.optimizer.add_param_group({"params": [c.log_var for c in self.criterion], "lr": lr})
        self.rgbseq = [0, 1, 2]
        self._rgb_idx = torch.tensor(self.rgbseq, device=device)      # (a Python list index would be a host-to-device copy per step: not capturable)
        self.reducer = GradReducer(self.params(), world, timing=timing) if (world > 1 and overlap) else None
        self.steps = 0
        self.status_every = status_every
        self._graph = None
        self._static = None
        # pipelined step (see the class docstring): state of the feature prefetch
        self._pf = None              # _Prefetch of the batch the next call is expected to bring
        self._pipe_h16 = True        # element type of the prefetched features (set per step by _pipe_ok)
        self._next = None            # the batch announced by the running call
        self._trunk_stream = None
        self._pipe_hook = None
        self.pipelined_steps = 0     # steps that consumed prefetched features (bench / tests read it)

    def _pipe_ok(self, h16):
        """the prefetch computes what features_for_head would hand to this head on the fast trunk path: fp16 channels_last features for the
        fp16-operand head (`self._pipe_h16`), fp32 ones for the exact-fp32 head"""
        ok = (not self.use_graph and os.environ.get("SRBH_TRAIN_PIPELINE", "1") == "1"
              and isinstance(getattr(self.net, "hrfeat", None), self._H.HRfeature)
              and hasattr(self.net_hr, "_use_strict") and not self.net_hr._use_strict() and not getattr(self.net_hr, "_train_path", False))
        self._pipe_h16 = bool(ok and FEATURE_H16 and h16 and os.environ.get("SRBH_PTAIL", "1") != "0")
        return ok

    def _launch_prefetch(self, nb):
        lr = nb[0]
        dev = lr.device
        if self._trunk_stream is None:
            self._trunk_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("SRBH_PIPE_PRIO", "0")))
        sT, cur = self._trunk_stream, torch.cuda.current_stream(dev)
        ev = torch.cuda.Event()
        ev.record(cur)
        sT.wait_event(ev)
        with torch.cuda.stream(sT), torch.no_grad():
            x3 = lr.index_select(1, self._rgb_idx)
            B = x3.shape[0]
            cap = pipe_images(dev)
            n = -(-B // cap)
            odt = torch.float16 if self._pipe_h16 else torch.float32
            fea = torch.empty((B, 64, 4 * x3.shape[2], 4 * x3.shape[3]), dtype=odt, device=dev, memory_format=torch.channels_last)
            # even split, in multiples of 8 images where the cap allows (the trunk's tile map keeps the row blocks of an image on one XCD
            # when the launch has a multiple of 8 workgroups per row block index, i.e. whole images per XCD: 64 -> 24, 24, 16 under a cap of 24)
            per = -(-B // n)
            if cap >= 8:
                per = min(cap // 8 * 8, -(-per // 8) * 8)
            # the tail convs (persistent, a whole CU's LDS per workgroup) on 3/4 of the CUs: a full grid stalls the main stream's chain
            # for its whole duration (same-box: 28.4 +- 2 ms with full grids, 27.0-27.6 with 192 of 256, profiles/r05ao_ab_ptail_wgs.txt)
            L = _lib.lib()
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            prev = L.srbh_ptail_wgs_cap(int(os.environ.get("SRBH_PIPE_TAIL_WGS", ncu * 3 // 4)))
            try:
                with self.net_hr.same_weights():          # (one walk over the 702 parameters for the launches of this prefetch, not one each)
                    i = 0
                    while i < B:
                        j = min(B, i + per)
                        self.net_hr.forward_feature(x3[i:j], out=fea[i:j], out_dtype=odt)
                        i = j
            finally:
                L.srbh_ptail_wgs_cap(prev)
            done = torch.cuda.Event()
            done.record(sT)
        lr.record_stream(sT)
        return _Prefetch(lr, fea, done)

    def _on_head_backward_done(self, _param):
        nb = self._next
        if nb is not None and self._pf is None:
            self._pf = self._launch_prefetch(nb)

    def params(self):
        return [p for g in self.optimizer.param_groups for p in g["params"]]

    def __call__(self, batch, next_batch=None):
        self._next = next_batch
        with self._H.head_precision(self.head_precision):
            if self.use_graph:
                return self._graph_step(batch)
            return self._step(batch)

    def _graph_step(self, batch):
        if self._graph is None:
            # static inputs: the floating-point tensors are views of ONE flat buffer, so that a new batch reaches them with a
            # single `cat` launch (+ one for the int64 labels) in front of the replay
            if getattr(self, "_static", None) is None:
                fl = [t for t in batch if t.dtype == torch.float32]
                flat = torch.empty(sum(t.numel() for t in fl), dtype=torch.float32, device=fl[0].device)
                views, o = [], 0
                for t in batch:
                    if t.dtype == torch.float32:
                        views.append(flat[o:o + t.numel()].view(t.shape))
                        o += t.numel()
                    else:
                        views.append(torch.empty_like(t))
                self._static, self._static_flat = tuple(views), flat
            self._stage(batch)
            if self.steps < 3:                       # eager warm-up (lazy packs, MIOpen kernels, Adam state) before the capture
                return self._step(self._static)
            self.net_hr.check_status()
            self.optimizer.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            # the graph OWNS the eager-made buffers it bakes in (the frozen RRDBNet's layer table / packed weights / workspace:
            # pinned, so calling net_hr at other geometries between replays cannot free them) -- wcache.Holder
            self._holder = wcache.Holder()
            with wcache.capturing(self._holder):
                h16 = self._H.head_h16()
                with torch.no_grad():
                    features_for_head(self.net_hr, self._static[0].index_select(1, self._rgb_idx), h16, model=self.net)   # eager: reports packs + workspace
                torch.cuda.synchronize()
                if self.reducer is not None:
                    self.reducer.paused = True
                try:
                    with torch.cuda.graph(self._graph):
                        self._static_out = self._step(self._static, in_graph=True)
                finally:
                    if self.reducer is not None:
                        self.reducer.paused = False
            self._stamped = [p for p in self.params()] + [b for b in self.net.buffers()]
        self._stage(batch)
        self._graph.replay()
        if self.world > 1:            # the replay left this rank's gradients in the captured .grad tensors
            self._reduce_and_update()
        # a replay updates the weights and the BatchNorm running statistics on the device without any Python-side trace (no
        # _version bump, no optimizer hook): stamp them, or eval / predict_tiles after training reuses the packed weights, folded
        # BatchNorm affines and captured predict graph of an earlier state (round-2 ADVICE)
        wcache.stamp(self._stamped)
        self.steps += 1
        if self.status_every and self.steps % self.status_every == 0 and hasattr(self.net_hr, "check_status"):
            self.net_hr.check_status()
        return self._static_out

    def _stage(self, batch):
        if all(d.data_ptr() == s.data_ptr() for d, s in zip(self._static, batch)):
            return                                   # the caller filled `static_batch()` in place
        for i, (d, s) in enumerate(zip(self._static, batch)):
            if tuple(d.shape) != tuple(s.shape) or d.dtype != s.dtype or d.device != s.device:
                # (torch.cat(out=) would silently RESIZE the flat buffer: the captured graph keeps reading the old views)
                raise ValueError(f"TrainStep(graph=True): batch tensor {i} is {tuple(s.shape)} {s.dtype} on {s.device}, the captured "
                                 f"step reads {tuple(d.shape)} {d.dtype} on {d.device}; a graph replays ONE batch geometry "
                                 "(drop the ragged last batch as the reference's loader does, train.py:97, or use graph=False)")
        torch.cat([t.reshape(-1) for t in batch if t.dtype == torch.float32], out=self._static_flat)
        for d, s in zip(self._static, batch):
            if d.dtype != torch.float32:
                torch.add(s, 0, out=d)

    def _reduce_and_update(self):
        if self.reducer is not None:
            self.reducer.finish()      # (buckets whose hooks did not fire -- all of them after a replay -- are launched here)
        else:
            allreduce_grads(self.params(), self.world)
        self.optimizer.step()

    def static_batch(self):
        """graph mode: the tensors the captured step reads (None before the first call); a loader may fill them in place"""
        return getattr(self, "_static", None)

    def _step(self, batch, in_graph=False):
        lr, height, height_aggre, build, weight, weight_aggre = batch
        h16 = self._H.head_h16()              # (in the grad mode the head runs in: 'auto' trains exact-fp32 and takes fp32 features)
        pipe = not in_graph and lr.is_cuda and self._pipe_ok(h16)
        pf, self._pf = self._pf, None
        if pipe and self._next is not None and self._pipe_hook is None:
            # SRBH_PIPE_AT=reg: the trunk goes out one phase earlier (when backward leaves `reg`, beside HRfeature's backward as well) -- A/B aid
            first = self.net.reg.fuse[0].conv1.weight if os.environ.get("SRBH_PIPE_AT", "hrfeat") == "reg" else self.net.hrfeat[0].conv1.weight
            self._pipe_hook = first.register_post_accumulate_grad_hook(self._on_head_backward_done)
        if not pipe:
            self._next = None
        if pf is not None and pipe and pf.matches(lr) and pf.fea.dtype == (torch.float16 if self._pipe_h16 else torch.float32):
            hr_fea = pf                       # a handle: the model issues the encoder / decoders first and waits in front of HRfeature
            self.pipelined_steps += 1
        else:
            if pf is not None and not in_graph:
                pf.abandon()
            with torch.no_grad():
                hr_fea = features_for_head(self.net_hr, lr.index_select(1, self._rgb_idx), h16, model=self.net)
        height_pred, build_pred, height_pred_aggre = self.net(lr, hr_fea)
        loss = (self.criterion[0](height_pred.squeeze(1), height, weight)
                + self.criterion[1](height_pred_aggre.squeeze(1), height_aggre, weight_aggre)
                + self.criterion[2](build_pred, build, weight))
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self._next is not None and self._pf is None:      # (the hook did not fire: no gradient reached HRfeature's first weight)
            self._pf = self._launch_prefetch(self._next)
        self._next = None
        if in_graph and self.world > 1:
            return loss.detach(), height_pred.detach()      # (the collectives and Adam follow each replay: _reduce_and_update)
        self._reduce_and_update()
        if in_graph:
            return loss.detach(), height_pred.detach()
        self.steps += 1
        # a persistent-trunk timeout NaN-poisons hr_fea (loud by itself); the host-side check names the reason.  It costs a
        # stream synchronisation, so: after the first step (where a co-residency problem shows) and then rarely.
        if self.steps == 1 or (self.status_every and self.steps % self.status_every == 0):
            if hasattr(self.net_hr, "check_status"):
                self.net_hr.check_status()
        return loss.detach(), height_pred.detach()


def synthetic_batch_device(batch, gen, device, aggregate=None):
    """`synthetic_batch` with every tensor drawn ON the device from the torch.Generator `gen` (device RNG): the epoch
    driver below must not be limited by host-side synthesis or H2D copies (the reference needs 8 loader workers for 20
    tiles/s).  Same shapes / label statistics as `synthetic_batch`."""
    lr = torch.rand((batch, 8, 64, 64), generator=gen, device=device)
    coarse = torch.rand((batch, 1, 32, 32), generator=gen, device=device)
    hval = torch.rand((batch, 1, 32, 32), generator=gen, device=device) ** 3 * 120.0
    height = torch.where(coarse > 0.82, hval, torch.zeros_like(hval))
    height = F.interpolate(height, scale_factor=8, mode="nearest").round()
    edges, class_weight = _label_consts(device)
    build = torch.bucketize(height[:, 0], edges, right=True)
    build = torch.where(height[:, 0] <= 0, torch.zeros_like(build), build).long().clamp_(0, 6)
    weight = class_weight[build]
    if aggregate is None:
        from .aggregate import aggregate_torch as aggregate
    height_aggre = aggregate(height, 0.25).reshape(batch, 64, 64)
    weight_aggre = aggregate(weight[:, None].contiguous(), 0.25).reshape(batch, 64, 64)
    return lr, height[:, 0], height_aggre, build, weight, weight_aggre


_LABEL_CONSTS = {}


def _label_consts(device):
    """(class edges, class weights) on the device, made once: a torch.tensor(list, device=...) per step is a pageable host-to-device
    copy that stalls the issuing thread -- inside the epoch loop that is the thread queueing the training step's launches"""
    key = str(torch.device(device))
    c = _LABEL_CONSTS.get(key)
    if c is None:
        c = _LABEL_CONSTS[key] = (torch.tensor(HIR[1:-1], dtype=torch.float32, device=device), torch.tensor(CLASS_WEIGHT, device=device))
    return c


def learnable_batch_device(batch, gen, device, aggregate=None):
    """`synthetic_batch_device` with labels that are a FUNCTION of the tile (the stock synthetic labels are independent of it, so a
    network can only learn their mean): height = blocky 8x8-HR-pixel map of the 2x2-LR-pixel mean of the three RRDB input channels,
    thresholded so that ~82 % of the pixels are ground (the reference's label statistics, bh_stats_globe.csv:2).  Used by the
    strict-vs-mixed convergence A/B (tools/convergence_ab.py, tests/test_gpu_convergence_ab.py): there the loss has to FALL for a
    reason, and how far it falls can be compared between precision modes."""
    lr = torch.rand((batch, 8, 64, 64), generator=gen, device=device)
    v = F.avg_pool2d(lr[:, :3].mean(1, keepdim=True), 2)                  # (B,1,32,32): mean of 12 uniforms, std 0.0833
    z = (v - 0.5) / 0.08333
    height = torch.where(z > 0.915, (z - 0.915) * 40.0 + 3.0, torch.zeros_like(z)).clamp_(0, 255)
    height = F.interpolate(height, scale_factor=8, mode="nearest").round()
    edges, class_weight = _label_consts(device)
    build = torch.bucketize(height[:, 0], edges, right=True)
    build = torch.where(height[:, 0] <= 0, torch.zeros_like(build), build).long().clamp_(0, 6)
    weight = class_weight[build]
    if aggregate is None:
        from .aggregate import aggregate_torch as aggregate
    height_aggre = aggregate(height, 0.25).reshape(batch, 64, 64)
    weight_aggre = aggregate(weight[:, None].contiguous(), 0.25).reshape(batch, 64, 64)
    return lr, height[:, 0], height_aggre, build, weight, weight_aggre


def train_epoch(ts, n_tiles, batch, rank, world, device, seed=1337, max_steps=None, aggregate=None):
    """BASELINE configs[3]: one data-parallel pass over `n_tiles` synthetic training tiles (31 500 = 45 000 x 0.7,
    data/datalist_globe_train_0.7.csv): every rank draws its own `batch` tiles per step on the device (seed + rank +
    step: no host synthesis, no H2D), drop_last like the reference's loader (train.py:97).  Returns (steps, tiles seen
    by the whole job, last loss)."""
    steps = n_tiles // (batch * world)
    if max_steps is not None:
        steps = min(steps, max_steps)
    gen = torch.Generator(device=device)
    loss = None

    def draw(i):
        gen.manual_seed(seed + 7919 * rank + 104729 * i)
        return synthetic_batch_device(batch, gen, device, aggregate)

    # the batch of step i + 1 is drawn BEFORE step i runs (what a loader's prefetch queue does) so that the pipelined TrainStep can
    # compute its RRDBNet features beside step i's backward
    nxt = draw(0) if steps else None
    for i in range(steps):
        cur, nxt = nxt, (draw(i + 1) if i + 1 < steps else None)
        loss, _ = ts(cur, next_batch=nxt)
    return steps, steps * batch * world, loss


PREDICT_GRAPH = os.environ.get("SRBH_PREDICT_GRAPH", "1") == "1"
PREDICT_SPLIT = os.environ.get("SRBH_PREDICT_SPLIT", "1") == "1"      # encoder / decoders as their own graph on a second stream (0: one graph, A/B aid)
# the encoder / decoders of batch k + 1 run BEHIND batch k's trunk (beside its tail convs / HRfeature / reg / seg) instead of beside batch
# k + 1's trunk (round 6; 0: the round-4 arrangement, A/B aid).  Measured (tools/predict_parts.py, profiles/r06x_*): the persistent trunk
# owns every CU, so the encoder's ~190 launches beside it are not hidden -- a batch takes the SUM of its three graphs -- and each one that
# slips in between two trunk launches delays a whole launch: 44.65 ms per 256 tiles against 43.7 with the encoder kept off the trunk.
PREDICT_AHEAD = os.environ.get("SRBH_PREDICT_AHEAD", "1") == "1"


class _PredictGraph:
    """One full batch of the tiled prediction -- RRDBNet features, encoder / decoders, HR head -- captured into a HIP graph: the
    ~700 launches of a batch (most of them the stock-op encoder's, a few microseconds of GPU work each) are issued by one
    hipGraphLaunch instead of by the Python host.  Bound to the networks' weights at capture time (the packed-weight buffers
    are baked into the graph): `key` changes when any parameter does, and predict_tiles re-captures."""

    def __init__(self, net_hr, model, batch, dev, chans):
        self.key = _PredictGraph.weights_key(net_hr, model, batch, dev)
        self.x = torch.zeros((batch, chans, 64, 64), device=dev)
        # The graph OWNS every eager-made device buffer it bakes in (wcache.Holder): RRDBNet's layer table / packed weights and
        # its (B, 64, 64) workspace -- pinned against the workspace budget until this object dies --, the head's weight packs, the
        # encoder's folded BatchNorm affines.  (Round 2 kept none of them: the second distinct ragged-tail size of a city run
        # evicted the B=128 workspace from a 2-entry LRU and every later replay wrote the trunk's activations into freed memory.)
        self.holder = wcache.Holder()
        with wcache.capturing(self.holder):
            side = torch.cuda.Stream(device=dev)      # warm-up off the capture (lazy packs, workspaces, MIOpen's solver search)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    model(self.x, features_for_head(net_hr, self.x[:, :3], model=model))
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            # THREE graphs, not one (round 4): the encoder / decoders (~250 small launches, 2-3 ms per 128 tiles) replay on a second
            # stream next to trunk + HRfeature on the main one, reg / seg follow the join.  Captured as ONE graph with a forked
            # stream the branches do not overlap on ROCm 7.2 (tools/graph_branch_probe.py: two forked chains replay slower than the
            # same chains on one stream); separate graph launches on separate streams do.  The persistent trunk owns every CU
            # while it runs, so the encoder fills in around it (before, between its launches, under HRfeature).
            self.split = PREDICT_SPLIT and all(hasattr(model, n) for n in ("forward_lr", "forward_hr", "forward_fuse"))
            self.ahead = self.split and PREDICT_AHEAD
            if self.ahead:
                # FIVE graphs + two sets of encoder buffers: trunk (its own 3-channel input), HRfeature, and per parity of the batch
                # number the encoder / decoders (own 8-channel input) and reg / seg reading that parity's decoder outputs
                self.side = side
                self.x3 = torch.zeros((batch, 3, 64, 64), device=dev)
                self.x_lr = [torch.zeros((batch, chans, 64, 64), device=dev) for _ in range(2)]
                self.g_trunk, self.g_hrfeat = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                self.g_lr2 = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]
                self.g_fuse2 = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]
                with torch.cuda.graph(self.g_trunk):
                    self.fea = features_for_head(net_hr, self.x3, model=model)
                with torch.cuda.graph(self.g_hrfeat):
                    self.hr_out = model.forward_hr(self.fea)
                self.lr_out2, self.out2 = [], []
                for p in range(2):
                    with torch.cuda.graph(self.g_lr2[p]):
                        self.lr_out2.append(model.forward_lr(self.x_lr[p]))
                    with torch.cuda.graph(self.g_fuse2[p]):
                        height, build = model.forward_fuse(self.lr_out2[p][0], self.lr_out2[p][1], self.hr_out)
                    self.out2.append((height, build) + ((self.lr_out2[p][2],) if self.lr_out2[p][2] is not None else ()))
                self.ev_lr = [torch.cuda.Event(), torch.cuda.Event()]
                self.ev_trunk = torch.cuda.Event()
                self.pending = [False, False]          # parity p's encoder / decoders were launched ahead (for the batch that comes next)
                self.out = self.out2[0]
                self.n = 0
            elif self.split:
                self.side = side
                self.g_lr, self.g_hr, self.g_fuse = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_lr):
                    self.lr_out = model.forward_lr(self.x)
                with torch.cuda.graph(self.g_hr):
                    self.hr_out = model.forward_hr(features_for_head(net_hr, self.x[:, :3], model=model))
                with torch.cuda.graph(self.g_fuse):
                    height, build = model.forward_fuse(self.lr_out[0], self.lr_out[1], self.hr_out)
                self.out = (height, build) + ((self.lr_out[2],) if self.lr_out[2] is not None else ())
            else:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.out = model(self.x, features_for_head(net_hr, self.x[:, :3], model=model))

    @staticmethod
    def weights_key(net_hr, model, batch, dev):
        """exact (no hashing: a collision would replay a stale graph): version, generation stamp and address of every parameter
        and buffer of both networks"""
        from . import hrfuse as _H
        k = tuple((t._version, getattr(t, "_srbh_gen", 0), t.data_ptr())
                  for m in (net_hr, model) for t in list(m.parameters()) + list(m.buffers()))
        return (k, batch, str(dev), wcache.gen(), _H._HEAD_PRECISION["mode"])

    def reset(self):
        """forget an encoder pass launched ahead (a city loop that ended early): the next call stages its own"""
        if self.ahead and any(self.pending):
            torch.cuda.current_stream(self.x.device).wait_stream(self.side)
            self.pending = [False, False]

    def _call_ahead(self, x_src, nxt_src):
        cur = torch.cuda.current_stream(self.x.device)
        p = self.n & 1
        if not self.pending[p]:                        # first batch of a run: this batch's encoder / decoders now, beside its trunk
            self.x_lr[p].copy_(x_src, non_blocking=True)
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self.g_lr2[p].replay()
                self.ev_lr[p].record(self.side)
        self.x3.copy_(x_src[:, :3], non_blocking=True)
        self.g_trunk.replay()
        if nxt_src is not None:                        # the next batch's encoder / decoders: behind this trunk, beside what follows it
            self.ev_trunk.record(cur)                  # (also behind reg / seg of batch k - 1, the last readers of that parity's buffers)
            self.side.wait_event(self.ev_trunk)
            with torch.cuda.stream(self.side):
                self.x_lr[1 - p].copy_(nxt_src, non_blocking=True)
                self.g_lr2[1 - p].replay()
                self.ev_lr[1 - p].record(self.side)
            self.pending[1 - p] = True
        self.g_hrfeat.replay()
        cur.wait_event(self.ev_lr[p])
        self.g_fuse2[p].replay()
        self.pending[p] = False
        self.n += 1
        return self.out2[p]

    def __call__(self, x_src, k, nxt_src=None):
        """one full batch; `nxt_src`: the tiles of the NEXT full batch of the same run (None: there is none)"""
        if self.ahead:
            return self._call_ahead(x_src, nxt_src)
        self.x[:k].copy_(x_src, non_blocking=True)
        if k < self.x.shape[0]:
            self.x[k:].zero_()
        if not self.split:
            self.graph.replay()
            return self.out
        cur = torch.cuda.current_stream(self.x.device)
        self.side.wait_stream(cur)                   # the batch is staged, the previous batch's reg / seg have read lr_out
        with torch.cuda.stream(self.side):
            self.g_lr.replay()
        self.g_hr.replay()
        cur.wait_stream(self.side)
        self.g_fuse.replay()
        return self.out


@torch.no_grad()
def predict_tiles(net_hr, model, tiles, posall, mosaic, batch=32, rank=0, world=1, pad_to=None):
    """predict_whole_image_grid's inner loop (predict_realesanet_feature_globe.py:167-185) for this rank's shard of a
    city's grid cells: RRDBNet feature extraction -> height / building heads -> quantise + integer mosaic on the
    device.  ``tiles`` (N,8,64,64) fp32, ``posall`` (N,4) LR-cell windows.  Merge shards with ``mosaic.all_reduce_``
    (or ``merge_``) before ``mosaic.finalize()``; integer sums make the result independent of the sharding.
    (Producing batch i+1's features on a second stream while batch i's heads run measured no gain: the persistent trunk
    kernel owns every CU while it runs, the launches simply serialise.)"""
    model.eval()
    net_hr.eval()
    lo, hi = shard_range(tiles.shape[0], rank, world)
    dev = mosaic.res_height.device
    pg = None
    if PREDICT_GRAPH and dev.type == "cuda" and hi - lo >= batch and hasattr(net_hr, "forward_feature"):
        pg = model.__dict__.get("_srbh_predict_graph")
        if pg is None or pg.key != _PredictGraph.weights_key(net_hr, model, batch, dev):
            pg = model.__dict__["_srbh_predict_graph"] = None      # (drop the old graph's memory pool first)
            pg = model.__dict__["_srbh_predict_graph"] = _PredictGraph(net_hr, model, batch, dev, tiles.shape[1])
    if pg is not None:
        pg.reset()
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        k = e - s
        if pg is not None and k == batch:
            out = pg(tiles[s:e], k, tiles[e:e + batch] if e + batch <= hi else None)
            mosaic.add(out[0], out[1], posall[s:e])
            continue
        x = tiles[s:e].to(dev, non_blocking=True)
        if k < batch:
            # ragged tail: run a padded batch (eval mode: tiles are independent) instead of a new tensor shape, for which the
            # stock-op encoder would first search / compile kernels (0.5 s per new shape, more than a small city's work).
            # pad_to = granularity of the padded sizes (None: always the full batch; e.g. 32 with batch 128 -> four shapes,
            # which the caller should have warmed up once)
            q = batch if not pad_to else min(batch, (k + pad_to - 1) // pad_to * pad_to)
            if q > k:
                x = torch.cat([x, x.new_zeros((q - k,) + tuple(x.shape[1:]))], 0)
        hr_fea = features_for_head(net_hr, x[:, :3], model=model)
        out = model(x, hr_fea)
        mosaic.add(out[0][:k], out[1][:k], posall[s:e])
    if hi > lo and hasattr(net_hr, "check_status"):
        # one synchronisation per city shard: a persistent-trunk timeout (another process holding CUs / LDS) already
        # NaN-poisoned the features; this raises with the reason instead of shipping a NaN mosaic
        net_hr.check_status()
    return hi - lo
