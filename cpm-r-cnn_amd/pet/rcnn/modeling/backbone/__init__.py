from .ResNet import *  # noqa: F401,F403
