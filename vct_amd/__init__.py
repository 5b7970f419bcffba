"""Importable alias of the product package.

The package directory is `video-captioning-transformer_amd/` (the repository's naming contract);
a hyphen cannot appear in a Python module name, so this shim exposes it as `vct_amd`: submodules
(`vct_amd.ops`, `vct_amd.model`, ...) resolve from that directory.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "video-captioning-transformer_amd")
__path__.insert(0, _REAL)
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
