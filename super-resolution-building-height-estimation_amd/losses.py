"""Loss modules of the training loop on libsrbh reductions.

Mirrors of reference losses_pytorch/selfloss.py with the same class names, constructor arguments, ``log_var`` parameter
and forward signatures: ``Dice`` (:6-17), ``MSE_adapt`` (:70-78), ``MSE_adapt_weight`` (:81-91), ``CE_DICE_adapt``
(:124-143), ``CE_DICE_adapt_weight`` (:145-168).  The full-resolution passes (squared error, log-softmax, softmax,
foreground probability, the four Dice sums, and their gradients) run in the HIP kernels of csrc/srbh_loss.hip; the
scalar arithmetic on the sums (mean, Dice ratio, exp(-log_var) weighting) stays in torch so ``log_var`` is trained by
autograd exactly as in the reference.  Unlike the reference the modules do not hard-code ``device="cuda"`` (:74,84,128,
151): ``log_var`` is created on ``device`` (default: the current CUDA device, which is what the reference does).

There is no CPU implementation: inputs must be CUDA fp32 tensors and libsrbh must be loadable."""
import torch
import torch.nn as nn

from . import _lib


def _dev(device):
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _f32c(t):
    if not t.is_cuda:
        raise RuntimeError("srbh losses run on the GPU only (got a %s tensor)" % t.device)
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


class _WMSESum(torch.autograd.Function):
    """sum_i w_i (pred_i - target_i)^2 as a 0-d fp32 tensor."""

    @staticmethod
    def forward(ctx, pred, target, weight):
        p, t = _f32c(pred), _f32c(target)
        if p.numel() != t.numel():
            raise RuntimeError(f"pred {tuple(pred.shape)} and target {tuple(target.shape)} differ in size")
        w = None
        if weight is not None:
            w = _f32c(weight.expand_as(p) if weight.shape != p.shape else weight)
        out = torch.zeros(1, dtype=torch.float64, device=p.device)
        L = _lib.lib()
        _lib.check(L.srbh_wmse_sum(p.data_ptr(), t.data_ptr(), w.data_ptr() if w is not None else None, p.numel(),
                                   out.data_ptr(), _lib.stream_ptr()), "srbh_wmse_sum")
        ctx.save_for_backward(p, t, w if w is not None else p.new_empty(0))
        ctx.has_w = w is not None
        ctx.shape = pred.shape
        return out[0].float()

    @staticmethod
    def backward(ctx, go):
        p, t, w = ctx.saved_tensors
        grad = torch.empty_like(p)
        g = go.reshape(1).float().contiguous()
        L = _lib.lib()
        _lib.check(L.srbh_wmse_grad(p.data_ptr(), t.data_ptr(), w.data_ptr() if ctx.has_w else None, p.numel(),
                                    g.data_ptr(), grad.data_ptr(), _lib.stream_ptr()), "srbh_wmse_grad")
        return grad.reshape(ctx.shape), None, None


class _CEDiceSums(torch.autograd.Function):
    """[sum w*CE, sum pb*tb, sum pb, sum tb] as a (4,) fp32 tensor; logits (B,C,H,W) in any dense layout."""

    @staticmethod
    def forward(ctx, logits, labels, weight):
        if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() < 2:
            raise RuntimeError("CE/Dice: logits must be a CUDA fp32 (B,C,...) tensor")
        z = logits
        if not (z.is_contiguous() or z.is_contiguous(memory_format=torch.channels_last)):
            z = z.contiguous()
        B, C = z.shape[0], z.shape[1]
        HW = z[0, 0].numel()
        if z.dim() == 4 and z.is_contiguous(memory_format=torch.channels_last) and not z.is_contiguous():
            bs, cs, ps = z.stride(0), 1, C
        else:
            bs, cs, ps = C * HW, HW, 1
        y = labels.contiguous()
        if y.dtype != torch.int64:
            y = y.long()
        if y.numel() != B * HW:
            raise RuntimeError(f"labels {tuple(labels.shape)} do not match logits {tuple(logits.shape)}")
        w = _f32c(weight) if weight is not None else None
        out = torch.zeros(4, dtype=torch.float64, device=z.device)
        L = _lib.lib()
        _lib.check(L.srbh_cedice_sums(z.data_ptr(), B, C, HW, bs, cs, ps, y.data_ptr(),
                                      w.data_ptr() if w is not None else None, out.data_ptr(), _lib.stream_ptr()),
                   "srbh_cedice_sums")
        ctx.save_for_backward(z, y, w if w is not None else z.new_empty(0))
        ctx.geo = (B, C, HW, bs, cs, ps, w is not None)
        return out.float()

    @staticmethod
    def backward(ctx, go):
        z, y, w = ctx.saved_tensors
        B, C, HW, bs, cs, ps, has_w = ctx.geo
        dz = torch.empty_like(z)            # preserves the (dense) memory format of z
        g = go.float().contiguous()
        L = _lib.lib()
        _lib.check(L.srbh_cedice_grad(z.data_ptr(), B, C, HW, bs, cs, ps, y.data_ptr(), w.data_ptr() if has_w else None,
                                      g.data_ptr(), dz.data_ptr(), _lib.stream_ptr()), "srbh_cedice_grad")
        return dz, None, None


class Dice(nn.Module):
    """selfloss.py:6-17 (plain torch: two reductions on an already materialised probability map)."""

    def forward(self, pred, target):
        smooth = 1.0
        n = pred.size(0)
        m1, m2 = pred.reshape(n, -1), target.reshape(n, -1)
        return 1 - (2.0 * (m1 * m2).sum() + smooth) / (m1.sum() + m2.sum() + smooth)


class _Adapt(nn.Module):
    def __init__(self, log_var=0.0, device=None):
        super().__init__()
        self.log_var = nn.Parameter(torch.tensor(float(log_var), device=_dev(device)))

    def _adapt(self, loss):
        return loss * torch.exp(-self.log_var) + self.log_var


class MSE_adapt_weight(_Adapt):
    """selfloss.py:81-91: mean(w * (x - t)^2) * exp(-log_var) + log_var."""

    def forward(self, inputs, targets, weight):
        return self._adapt(_WMSESum.apply(inputs, targets, weight) / inputs.numel())


class MSE_adapt(_Adapt):
    """selfloss.py:70-78."""

    def forward(self, inputs, targets):
        return self._adapt(_WMSESum.apply(inputs, targets, None) / inputs.numel())


class CE_DICE_adapt_weight(_Adapt):
    """selfloss.py:145-168: mean(w * CE) + Dice(softmax[:,1:].sum(1), rmask > 0), uncertainty-weighted."""

    def forward(self, pmask, rmask, weight):
        s = _CEDiceSums.apply(pmask, rmask, weight)
        loss_ce = s[0] / rmask.numel()
        loss_dice = 1 - (2.0 * s[1] + 1.0) / (s[2] + s[3] + 1.0)
        return self._adapt(loss_ce + loss_dice)


class CE_DICE_adapt(_Adapt):
    """selfloss.py:124-143."""

    def forward(self, pmask, rmask):
        s = _CEDiceSums.apply(pmask, rmask, None)
        loss_ce = s[0] / rmask.numel()
        loss_dice = 1 - (2.0 * s[1] + 1.0) / (s[2] + s[3] + 1.0)
        return self._adapt(loss_ce + loss_dice)
