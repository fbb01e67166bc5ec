from .grid_heads import *  # noqa: F401,F403
from .cls_heads import *  # noqa: F401,F403
