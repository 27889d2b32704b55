from .mcgan import mcgan, MCGAN  # noqa: F401
from . import utils  # noqa: F401
from .mcglow import mcglow, MCGlow  # noqa: F401
from .mcpixelcnn import mcpixelcnn, MCGatedPixelCNN  # noqa: F401
from .mcvae import mcvae, MCVAE  # noqa: F401
from .vqvae import vqvae, VQVAE  # noqa: F401
from .classifier import classifier, Classifier  # noqa: F401
