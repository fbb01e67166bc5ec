from .FPN import *  # noqa: F401,F403
