"""Drop-in for the reference's src/models/__init__.py: `import models; models.mcgan()` (train_gan.py:3,76) builds
the MI355X module trees -- same factories, class names and state_dict keys (models/mcgan.py, mcvae.py, mcglow.py,
mcpixelcnn.py, vqvae.py, utils.py).  The non-MC baselines (cgan, cvae, cglow, cpixelcnn) carry no
MultimodalController op and stay the reference's own files."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import _path  # noqa: F401,E402
from mcgen_amd.models.mcgan import *  # noqa: F401,F403,E402
from mcgen_amd.models.mcglow import *  # noqa: F401,F403,E402
from mcgen_amd.models.mcpixelcnn import *  # noqa: F401,F403,E402
from mcgen_amd.models.mcvae import *  # noqa: F401,F403,E402
from mcgen_amd.models.vqvae import *  # noqa: F401,F403,E402
from mcgen_amd.models import utils  # noqa: F401,E402
from mcgen_amd.models.mcgan import mcgan  # noqa: F401,E402
from mcgen_amd.models.mcglow import mcglow  # noqa: F401,E402
from mcgen_amd.models.mcpixelcnn import mcpixelcnn  # noqa: F401,E402
from mcgen_amd.models.mcvae import mcvae  # noqa: F401,E402
from mcgen_amd.models.vqvae import vqvae  # noqa: F401,E402
from mcgen_amd.models.classifier import classifier, Classifier  # noqa: F401,E402
