"""MI355X-native hot path of lauraset/Super-resolution-building-height-estimation.

The directory name follows the build contract (``<repo-name>_amd``) and is not a valid Python
identifier, so it is imported through the alias package ``srbh_amd`` (see ``srbh_amd.py`` at the
repo root), e.g. ``from srbh_amd.rrdbnet import RRDBNet``.  The reference's own import paths
(``SR.rrdbnet_arch``, ``SR.HRfuse``, ``mymodels``, ``aggregate_utils``) are provided by the shim
directory ``dropin/`` (put it on ``sys.path`` ahead of the reference checkout).
"""
__version__ = "0.1.0"
