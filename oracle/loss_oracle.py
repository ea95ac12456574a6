"""CPU oracle of the loss and metric arithmetic  --  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline).

Functional torch restatement of reference losses_pytorch/selfloss.py (Dice :6-17, MSE_adapt :70-78, MSE_adapt_weight
:81-91, CE_DICE_adapt :124-143, CE_DICE_adapt_weight :145-168) and metrics.py (SegmentationMetric.genConfusionMatrix
:67-74, HeightMetric.addBatch :186-200 and its getters :202-215).  Pinned against the imported reference by
tools/make_golden.py (fixtures tests/golden/g10_losses.npz, g11_metrics.npz; the reference's own toy vectors at
metrics.py:466-469 are part of g11)."""
import torch
import torch.nn.functional as F


def dice(pred, target):
    smooth = 1.0                                                  # selfloss.py:12
    n = pred.size(0)
    m1, m2 = pred.reshape(n, -1), target.reshape(n, -1)           # :14-15
    inter = (m1 * m2).sum()                                       # :16
    return 1 - (2.0 * inter + smooth) / (m1.sum() + m2.sum() + smooth)   # :17


def mse_adapt_weight(inputs, targets, weight, log_var):
    loss = F.mse_loss(inputs, targets, reduction="none")         # :87
    loss = (loss * weight).mean() if weight is not None else loss.mean()   # :88 (:75 unweighted)
    return loss * torch.exp(-log_var) + log_var                   # :89-90


def ce_dice_adapt_weight(pmask, rmask, weight, log_var):
    ce = F.cross_entropy(pmask, rmask, reduction="none")         # :149,157
    ce = (ce * weight).mean() if weight is not None else ce.mean()   # :158 (:136 unweighted)
    fg = pmask.softmax(dim=1)[:, 1:].sum(dim=1)                   # :160-161
    loss = ce + dice(fg, (rmask > 0))                             # :162-164
    return loss * torch.exp(-log_var) + log_var                   # :165-166


def confusion_matrix(pred, label, num_class):
    idx = num_class * label.flatten() + pred.flatten()            # metrics.py:71
    return torch.bincount(idx, minlength=num_class ** 2).reshape(num_class, num_class)   # :72-73


def height_metric_batch(pred, ref, cls, num_class):
    """One addBatch (metrics.py:186-200): returns the (num_class,3) float64 increments of stats and the (num_class,1)
    increments of count."""
    stats = torch.zeros((num_class, 3), dtype=torch.float64)
    count = torch.zeros((num_class, 1), dtype=torch.float64)
    for i in range(num_class):
        mask = cls == i
        c = mask.sum().float()
        if int(c.item()) == 0:
            continue
        d = pred[mask] - ref[mask]
        stats[i, 0] += torch.sqrt((d ** 2).mean()) * c            # rmse * count (:193,197)
        stats[i, 1] += d.abs().mean() * c                         # :194,198
        stats[i, 2] += d.mean() * c                               # :195,199
        count[i] += c
    return stats, count
