"""MI355X-native disparity hot path of autonomousvision/connecting_the_dots.

`connecting_the_dots_amd.torchext` mirrors the reference's `torchext.functions` /
`torchext.modules` API (torchext/functions.py, torchext/modules.py); the arithmetic runs in
hand-written HIP kernels (csrc/) behind the C ABI of include/ctd_hip.h.
"""
from . import torchext  # noqa: F401

__version__ = "0.1.0"
