"""Import alias for the package directory
``multimodal-controller-for-generative-models_amd`` (its name is not a valid
Python identifier).  ``import mcgen_amd`` == that package."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      'multimodal-controller-for-generative-models_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _f
