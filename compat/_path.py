"""Makes the repo root importable from the shim files (they sit where the reference's `src/` files sit)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
