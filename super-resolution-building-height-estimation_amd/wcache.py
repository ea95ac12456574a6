"""Generation counter of the packed-weight caches.

libsrbh's kernels read convolution weights in kernel-specific packed layouts (HWPACK32, the trunk's layer table, transposed /
flipped packs for the data gradients, folded BatchNorm affines); the Python modules cache those packs per parameter.  A cache key of
`(param._version, param.data_ptr())` alone is NOT safe: `torch.optim.Adam(fused=True)` -- and any other multi-tensor fused
optimizer -- updates the parameters without touching their version counters (checked on torch 2.10: `_version` stays put), so a
training loop with a fused optimizer would keep convolving with the weights of step 0.  Every cache key therefore also carries
`gen(param)`: a global optimizer post-step hook stamps the parameters of the optimizer that just stepped (only those: the frozen
RRDBNet of the height stage keeps its packs while the regression network trains).  Code that writes parameters behind
autograd's back (`p.data.copy_(...)`, custom CUDA/HIP updates) calls `invalidate_weight_caches()` itself."""
import torch

_STEP = [0]
_MANUAL = [0]


def gen(*params):
    """cache-key component: changes whenever an optimizer has stepped one of `params` (or after invalidate_weight_caches())"""
    return tuple(getattr(p, "_srbh_gen", 0) for p in params if p is not None) + (_MANUAL[0],)


def invalidate_weight_caches() -> None:
    _MANUAL[0] += 1


def stamp(tensors) -> None:
    """Mark `tensors` (parameters OR buffers) as changed behind the version counters: what the optimizer hook does for the
    parameters an optimizer stepped.  `harness.TrainStep(graph=True)` calls it after every graph replay for the parameters
    AND the BatchNorm buffers of the trained network -- a replay updates both on the device without any Python-side trace."""
    _STEP[0] += 1
    g = _STEP[0]
    for t in tensors:
        if t is not None:
            t._srbh_gen = g


def _after_step(optimizer, args, kwargs):
    stamp(p for group in optimizer.param_groups for p in group["params"])
    import sys
    hr = sys.modules.get(__package__ + ".hrfuse")       # (only if the head module is in use: its registered 16-bit packs are refreshed in one launch)
    if hr is not None:
        hr.repack_after_step(optimizer)


# ---- buffers a captured HIP graph points at ---------------------------------------------------------------------------
# A captured graph bakes device ADDRESSES in: packed-weight buffers, the trunk's layer table, workspaces.  Those live in Python
# caches that replace (and thereby free) their buffers whenever a key changes or an LRU evicts -- while the graph keeps
# replaying into the freed memory (round-2 VERDICT: the 2-entry workspace LRU of RRDBNet under harness._PredictGraph).  Rule: a
# graph OWNS what it points at.  Code that captures wraps warm-up + capture in `with wcache.capturing(holder)`; every cache
# that hands out a device buffer reports it through `keep(...)`, which appends it to the innermost active holder (a no-op
# otherwise); the holder lives exactly as long as the graph object.  Buffers allocated DURING the capture come from the graph's
# private pool and need no holder; the ones made by the eager warm-up do.
_HOLDERS = []


class Holder:
    """Python references (tensors, ctypes arrays, ...) a captured graph needs alive + un-pin callbacks run when it dies."""

    def __init__(self):
        self.refs = []
        self._unpin = []

    def on_release(self, fn):
        self._unpin.append(fn)

    def release(self):
        for fn in self._unpin:
            try:
                fn()
            except Exception:
                pass
        self._unpin = []
        self.refs = []

    def __del__(self):
        self.release()


class capturing:
    def __init__(self, holder: Holder):
        self.holder = holder

    def __enter__(self):
        _HOLDERS.append(self.holder)
        return self.holder

    def __exit__(self, *exc):
        _HOLDERS.pop()
        return False


def active_holder():
    return _HOLDERS[-1] if _HOLDERS else None


def keep(*objs) -> None:
    if _HOLDERS:
        _HOLDERS[-1].refs.extend(o for o in objs if o is not None)


from torch.optim.optimizer import register_optimizer_step_post_hook  # noqa: E402

_HOOK = register_optimizer_step_post_hook(_after_step)
