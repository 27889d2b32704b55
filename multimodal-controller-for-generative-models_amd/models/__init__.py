from .mcgan import mcgan, MCGAN  # noqa: F401
from . import utils  # noqa: F401
