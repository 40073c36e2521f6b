"""Mirror of the reference's `utils` package surface used on the hot path (utils/images.py __all__)."""
from .images import *  # noqa: F401,F403
