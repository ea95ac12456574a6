"""shim for the reference import path aggregate_utils."""
import _bootstrap  # noqa: F401
from srbh_amd.aggregate import aggregate_torch  # noqa: F401,E402
