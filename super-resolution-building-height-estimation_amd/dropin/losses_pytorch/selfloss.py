"""shim for the reference import path losses_pytorch.selfloss (train.py:20)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: F401,E402
from srbh_amd.losses import Dice, MSE_adapt, MSE_adapt_weight, CE_DICE_adapt, CE_DICE_adapt_weight  # noqa: F401,E402
