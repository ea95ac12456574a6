"""shim for the reference import path SR.HRfuse."""
from srbh_amd.hrfuse import (BasicBlock, GeoNet, HRfeature, HRfuse, HRfuse_residual, HRfuse_x2, HRupsample,  # noqa: F401
                             Refine_residual, Upsampler, conv1x1, conv3x3, default_conv)
