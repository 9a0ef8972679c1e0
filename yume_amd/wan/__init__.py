from .modules.model import WanModel  # noqa: F401
