"""wan23/modules/attention.py seam — see yume_amd/attention.py."""
from ...attention import attention, flash_attention  # noqa: F401
