"""shim for the reference import path metrics (train.py:13): the tensor reductions; the CSV dumpers stay with the caller."""
import _bootstrap  # noqa: F401
from srbh_amd.metrics import AverageMeter, SegmentationMetric, HeightMetric  # noqa: F401,E402
