"""Validation metrics on libsrbh reductions.

Mirrors of reference metrics.py: ``AverageMeter`` (:137-157), ``SegmentationMetric`` (:6-87: confusion matrix by
bincount, OA / precision / recall / F1 / IoU / FWIoU) and ``HeightMetric`` (:160-229: per-hierarchy-class RMSE, MAE, ME
weighted by the pixel count of each batch).  ``addBatch`` runs one HIP reduction per batch (csrc/srbh_loss.hip) instead
of a bincount plus, for HeightMetric, seven boolean-mask gathers with a host synchronisation each (``count.item()``,
:190); the accumulators stay float64 tensors on the device like the reference's.  GPU only."""
import torch
import torch.nn as nn

from . import _lib


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def _i64(t):
    t = t.contiguous()
    return t if t.dtype == torch.int64 else t.long()


class SegmentationMetric(nn.Module):
    def __init__(self, numClass, device="cuda"):
        super().__init__()
        self.numClass = numClass
        self.device = device
        self.reset()
        self.count = 0

    def reset(self):
        self.confusionMatrix = torch.zeros((self.numClass, self.numClass), dtype=torch.float64, device=self.device)
        self._bad = torch.zeros((), dtype=torch.int64, device=self.device)   # out-of-range label / prediction seen (sticky)

    def genConfusionMatrix(self, imgPredict, imgLabel):
        """metrics.py:67-74: cm[label, pred] counts as an int64 (numClass, numClass) tensor."""
        p, y = _i64(imgPredict), _i64(imgLabel)
        if not p.is_cuda:
            raise RuntimeError("srbh metrics run on the GPU only")
        cm = torch.zeros(self.numClass * self.numClass + 1, dtype=torch.int64, device=p.device)   # last word: range flag
        L = _lib.lib()
        _lib.check(L.srbh_confusion_add(p.data_ptr(), y.data_ptr(), p.numel(), self.numClass, cm.data_ptr(),
                                        cm.data_ptr() + 8 * self.numClass * self.numClass, _lib.stream_ptr()),
                   "srbh_confusion_add")
        return cm[:-1].reshape(self.numClass, self.numClass), cm[-1]

    def addBatch(self, imgPredict, imgLabel):
        assert imgPredict.shape == imgLabel.shape
        cm, bad = self.genConfusionMatrix(imgPredict, imgLabel)
        self.confusionMatrix += cm
        self._bad |= bad              # stays on the device: addBatch never synchronises

    def check_range(self):
        """Raise if any batch held a label / prediction outside [0, numClass): the reference's bincount
        (metrics.py:71-73) raises on negatives or fails to reshape, a kernel can only flag.  Called where the
        scores are read, which is a host synchronisation anyway."""
        if int(self._bad):
            raise ValueError("SegmentationMetric: a label or prediction outside [0, %d) was passed to addBatch" % self.numClass)

    def getConfusionMatrix(self):
        self.check_range()
        return self.confusionMatrix

    def OverallAccuracy(self):
        self.check_range()
        return torch.diag(self.confusionMatrix).sum() / self.confusionMatrix.sum()

    def Precision(self):
        return torch.diag(self.confusionMatrix) / self.confusionMatrix.sum(0)

    def Recall(self):
        return torch.diag(self.confusionMatrix) / self.confusionMatrix.sum(1)

    def F1score(self):
        p, r = self.Precision(), self.Recall()
        return 2 * p * r / (p + r)

    def IntersectionOverUnion(self):
        inter = torch.diag(self.confusionMatrix)
        union = self.confusionMatrix.sum(1) + self.confusionMatrix.sum(0) - inter
        return inter / union

    def meanIntersectionOverUnion(self):
        return torch.mean(self.IntersectionOverUnion())

    def Frequency_Weighted_Intersection_over_Union(self):
        freq = self.confusionMatrix.sum(1) / (self.confusionMatrix.sum() + 1e-8)
        inter = torch.diag(self.confusionMatrix)
        iu = inter / (self.confusionMatrix.sum(1) + self.confusionMatrix.sum(0) - inter + 1e-8)
        return freq * iu

    def mFWIoU(self):
        return self.Frequency_Weighted_Intersection_over_Union().sum()


class HeightMetric(nn.Module):
    def __init__(self, numClass=7, device="cuda"):
        super().__init__()
        self.numClass = numClass
        self.device = device
        self.reset()

    def reset(self):
        self.count = torch.zeros((self.numClass, 1), dtype=torch.float64, device=self.device)
        self.stats = torch.zeros((self.numClass, 3), dtype=torch.float64, device=self.device)   # rmse, mae, me
        self.balance_stats = torch.zeros((self.numClass, 3), dtype=torch.float64, device=self.device)

    def addBatch(self, pred, ref, buildhir):
        """metrics.py:186-200: per class, rmse/mae/me of THIS batch times its pixel count are accumulated."""
        p = pred.float().contiguous()
        r = ref.float().contiguous()
        c = _i64(buildhir)
        if not p.is_cuda:
            raise RuntimeError("srbh metrics run on the GPU only")
        s = torch.zeros((self.numClass, 4), dtype=torch.float64, device=p.device)
        L = _lib.lib()
        _lib.check(L.srbh_height_metric_sums(p.data_ptr(), r.data_ptr(), c.data_ptr(), p.numel(), self.numClass,
                                             s.data_ptr(), _lib.stream_ptr()), "srbh_height_metric_sums")
        n = s[:, 3:4]
        safe = n.clamp_min(1.0)
        self.stats[:, 0:1] += torch.sqrt(s[:, 0:1] / safe) * n      # classes without pixels contribute 0 (:190-191)
        self.stats[:, 1:2] += s[:, 1:2]                             # mae * count == sum |d|
        self.stats[:, 2:3] += s[:, 2:3]
        self.count += n

    def getAvgEach(self):
        return self.stats / (self.count + 1e-10)

    def getAvgBalance(self):
        return self.getAvgEach().mean(dim=0)

    def getAvgAll(self):
        return self.stats.sum(dim=0) / self.count.sum()

    def getCount(self):
        return self.count
