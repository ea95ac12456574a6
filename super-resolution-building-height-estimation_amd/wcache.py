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


def _after_step(optimizer, args, kwargs):
    _STEP[0] += 1
    g = _STEP[0]
    for group in optimizer.param_groups:
        for p in group["params"]:
            p._srbh_gen = g


from torch.optim.optimizer import register_optimizer_step_post_hook  # noqa: E402

_HOOK = register_optimizer_step_post_hook(_after_step)
