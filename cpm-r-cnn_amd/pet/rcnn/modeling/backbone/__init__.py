from .ResNet import *  # noqa: F401,F403
from .ResNeXt import *  # noqa: F401,F403
