"""Importable alias for the product package, whose directory name
(``collaborative-gan-sampling_amd``) is not a valid Python identifier.

``import cgs_amd`` exposes that directory as this package (same modules, one copy)."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "collaborative-gan-sampling_amd")
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
del _f
