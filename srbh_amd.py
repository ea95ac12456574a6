"""Import alias: ``srbh_amd`` -> the package directory ``super-resolution-building-height-estimation_amd/``
(whose name, fixed by the build contract, is not an importable identifier)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)),
                          "super-resolution-building-height-estimation_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
