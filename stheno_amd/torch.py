"""``import stheno_amd.torch as stheno`` -- the counterpart of ``stheno/torch.py``: the
only backend of this package is torch on ROCm, so this simply re-exports the package."""
from . import *  # noqa: F401,F403
from . import B  # noqa: F401
