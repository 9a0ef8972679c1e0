from .model import WanModel  # noqa: F401
from .attention import flash_attention  # noqa: F401
