from .fdd import *        # noqa: F401,F403
from .gp import *         # noqa: F401,F403
from .observations import *  # noqa: F401,F403
from .measure import *    # noqa: F401,F403
