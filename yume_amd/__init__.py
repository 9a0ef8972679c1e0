"""yume_amd — MI355X (gfx950) native implementation of YUME's denoise hot path.

    yume_amd.wan23.modules.model.WanModel   drop-in DiT, Yume-5B-720P architecture
    yume_amd.wan.modules.model.WanModel     drop-in DiT, Yume-I2V-14B-540P architecture
    yume_amd.attention.flash_attention      operator seam (head_dim 128, bf16)
    yume_amd.ops                            tensor-level wrappers of the C-ABI in include/yume_hip.h
"""
__version__ = "0.1.0"
