"""Alias of :mod:`raytracer_amd.rtsimdataloader` under the name the reference's rtscene.py:4 and
rttests.py:109 import (``rtsimdataloader_new``): the same RTMesh / SurfType / loaders."""
if not __package__:  # top-level import (this directory on sys.path): see rtmaincpp.py
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    if _root not in _sys.path:
        _sys.path.append(_root)
from raytracer_amd.rtsimdataloader import *  # noqa: E402,F401,F403
from raytracer_amd.rtsimdataloader import RTMesh, SurfType  # noqa: E402,F401
