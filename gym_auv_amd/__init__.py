"""Importable alias for the ``gym-auv_amd/`` source directory.

The product directory carries the reference's name (with its hyphen), which
Python cannot import directly; this package maps the import name
``gym_auv_amd`` onto that directory, so ``gym_auv_amd.env`` is
``gym-auv_amd/env.py``.
"""
import os as _os

_SRC = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gym-auv_amd")
__path__.insert(0, _SRC)
with open(_os.path.join(_SRC, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_SRC, "__init__.py"), "exec"))
del _f
