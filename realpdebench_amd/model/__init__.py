from .model import Model            # noqa: F401
from .load_model import load_model  # noqa: F401
