"""shim for the reference import path mymodels (the model train.py / predict instantiate)."""
import _bootstrap  # noqa: F401
from srbh_amd.models import SRRegress_Cls_feature  # noqa: F401,E402
