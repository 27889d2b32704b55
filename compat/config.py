"""Drop-in for the reference's src/config.py (config.py:1-5): `from config import cfg` binds the same dict the
MI355X package reads, with the reference's default keys (config.yml) and its process_control table."""
import _path  # noqa: F401
from mcgen_amd.config import cfg, process_control  # noqa: F401
