"""shim for the reference import path SR.rrdbnet_arch (hot-path symbols only)."""
from srbh_amd.rrdbnet import (RRDB, RRDBNet, RealESRGAN, ResidualDenseBlock, default_init_weights,  # noqa: F401
                              make_layer, pixel_unshuffle)
