"""Drop-in for the op half of the reference's `torchext` package (torchext/__init__.py:3-4)."""
from .functions import *  # noqa: F401,F403
from .modules import *  # noqa: F401,F403
