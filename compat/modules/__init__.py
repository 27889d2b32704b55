"""Drop-in for the reference's src/modules/__init__.py (modules/modules.py:6-85)."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import _path  # noqa: F401,E402
from mcgen_amd.modules import MultimodalController, Wrapper, VectorQuantization  # noqa: F401,E402
