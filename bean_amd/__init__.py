"""Importable alias for the ``crispr-bean_amd/`` package directory.

``crispr-bean_amd`` is not a valid Python identifier, so this shim exposes the
same sources as ``bean_amd`` by pointing the package search path at that
directory: ``bean_amd.model.run`` is ``crispr-bean_amd/model/run.py``.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_src = _os.path.join(_os.path.dirname(_here), "crispr-bean_amd")
if not _os.path.isdir(_src):  # pragma: no cover
    raise ImportError(f"bean_amd: source directory {_src} is missing")
__path__.append(_src)

__version__ = "0.1.0"
