from .trainer import *  # noqa: F401,F403
