# the reference exports the factory under the submodule's name (qtypes/__init__.py:1); the manager
# looks it up as qtypes.__dict__['int_quantizer'] (inference_quantization_manager.py:401-405)
from .int_quantizer import int_quantizer, IntQuantizer  # noqa: F401
from .dummy_quantizer import DummyQuantizer  # noqa: F401
