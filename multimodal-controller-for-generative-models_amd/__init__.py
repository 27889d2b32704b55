"""MI355X-native training path for MultimodalController generative models.

Python host side (nn.Module surface mirroring the reference's
``src/modules/modules.py`` and ``src/models/mcgan.py``) over the C ABI of
``csrc/libmcgen_hip.so`` (declared in ``include/mcgen_hip.h``).  There is no CPU
fallback: every compute entry point raises if the HIP library is missing or a
tensor is not on a ROCm device.
"""
__version__ = '0.1.0'
