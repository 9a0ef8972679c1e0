"""ctypes binding of libyume_hip.so (the C-ABI declared in include/yume_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this
module raises at import of the symbol, and every op raises RuntimeError on a non-zero return code.
"""
import ctypes
import os
from ctypes import c_void_p, c_int, c_int64, c_float, c_char_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# YUME_HIP_LIB: an experiment build of the same ABI (tools/build_variant.sh) instead of the product library — A/B runs of whole benches
LIB_PATH = os.environ.get("YUME_HIP_LIB") or os.path.join(_HERE, "lib", "libyume_hip.so")

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float

# name -> argtypes  (restype is int unless listed in _RES)
SIGNATURES = {
    "yume_last_error": [],
    "yume_abi_version": [],
    "yume_target_arch": [],
    "yume_counter_workspace_bytes": [],
    "yume_counter_workspace_init": [_P, _L, _P],
    "yume_calibrate_mfma": [_L, _L, _P, _P, _P],
    "yume_adaln_modulate": [_P, _L, _L, _L, _F, _P, _P, _L, _P, _I, _P, _L, _I, _P],
    "yume_gemm_bf16": [_P, _L, _P, _L, _P, _L, _L, _L, _I, _P, _L, _P, _L, _P, _P, _L, _L, _I, _P],
    "yume_gemm_workspace_bytes": [],
    "yume_gemm_bf16_ws": [_P, _L, _P, _L, _P, _L, _L, _L, _I, _P, _L, _P, _L, _P, _P, _L, _L, _I, _P, _L, _P],
    "yume_rmsnorm_f32": [_P, _L, _L, _L, _F, _P, _P, _L, _P],
    "yume_gemm_bf16_batched": [_P, _L, _L, _P, _L, _L, _L, _L, _L, _I, _P, _L, _L, _L, _I, _P],
    "yume_gemm_splitk_workspace_bytes": [_L, _L, _I],
    "yume_gemm_bf16_splitk": [_P, _L, _P, _L, _P, _L, _L, _L, _I, _P, _L, _I, _P, _P],
    "yume_softmax_bias_rows": [_P, _L, _L, _L, _L, _P, _L, _P, _L, _L, _P],
    "yume_rmsnorm_rows_periodic": [_P, _L, _L, _L, _P, _L, _F, _P],
    "yume_rmsnorm_rope": [_P, _L, _L, _L, _I, _P, _F, _P, _L, _P],
    "yume_attn_fwd": [_P, _L, _P, _L, _P, _L, _P, _L, _L, _L, _L, _F, _I, _I, _P],
    "yume_attn_workspace_bytes": [_L, _L, _L],
    "yume_attn_fwd_ws": [_P, _L, _P, _L, _P, _L, _P, _L, _L, _L, _L, _F, _I, _I, _P, _L, _P],
    "yume_linear_smallm_f32": [_P, _L, _L, _P, _I, _P, _L, _I, _I, _P, _P, _P],
    "yume_sinusoidal_embed": [_P, _P, _L, _L, _P, _P],
    "yume_modulation_table": [_P, _P, _L, _L, _L, _P, _P],
    "yume_patch_gather": [_P, _I, _L, _L, _L, _L, _L, _L, _L, _L, _P, _L, _P],
    "yume_unpatchify": [_P, _L, _L, _L, _L, _L, _L, _L, _P, _P],
    "yume_cast_bf16": [_P, _L, _L, _L, _L, _P, _L, _P],
    "yume_transpose_bf16": [_P, _I, _L, _L, _L, _P, _L, _P],
    "yume_conv3d_cl": [_P, _P, _L, _L, _L, _L, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _I, _P, _L,
                       _P, _L, _P, _P],
    "yume_vae_rmsnorm_silu": [_P, _L, _L, _L, _P, _P, _I, _P, _L, _P],
    "yume_vae_dupup_add": [_P, _L, _L, _L, _L, _L, _P, _L, _L, _L, _I, _I, _I, _P],
    "yume_vae_avgdown_add": [_P, _L, _L, _L, _L, _L, _P, _L, _L, _I, _I, _P],
    "yume_softmax_rows": [_P, _L, _L, _L, _F, _P, _L, _P],
    "yume_vae_pack_input": [_P, _I, _L, _L, _L, _L, _I, _P, _P, _P, _L, _P],
    "yume_frames_u8": [_P, _L, _L, _L, _L, _P, _P],
    "yume_frames_u8_trunc": [_P, _L, _L, _L, _L, _P, _P],
    "yume_vae_unpack_output": [_P, _L, _L, _L, _L, _L, _I, _P, _P, _F, _F, _P, _P],
}
_RES = {"yume_last_error": c_char_p, "yume_target_arch": c_char_p, "yume_gemm_splitk_workspace_bytes": c_int64, "yume_attn_workspace_bytes": c_int64,
        "yume_counter_workspace_bytes": c_int64, "yume_gemm_workspace_bytes": c_int64}

_lib = None
ABI_VERSION = 8          # must equal YUME_ABI_VERSION in include/yume_hip.h; bumped whenever an argument list changes


def load():
    """Load libyume_hip.so (once). Raises RuntimeError with a build hint if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `python -m yume_amd.build`). "
            "There is no CPU/PyTorch fallback for the yume_amd hot path.")
    # torch first: its bundled HIP runtime must be the one already in the process when this library's libamdhip64 dependency is
    # resolved (same SONAME, so the loader shares it). The other order puts two runtimes in one process and the second one finds
    # no device ("no ROCm-capable device is detected" on the first launch; seen with build() before `import torch`).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RES.get(name, c_int)
    # the .so is built out of band (git-ignored): a stale one with the same symbol names but other argument lists would be
    # called with the wrong ctypes signatures, so refuse anything but this binding's ABI and target
    got, arch = int(lib.yume_abi_version()), lib.yume_target_arch()
    if got != ABI_VERSION or arch != b"gfx950":
        raise RuntimeError(f"{LIB_PATH} has ABI version {got} for {arch!r}, this binding needs {ABI_VERSION} for gfx950: "
                           "rebuild with `python -m yume_amd.build --force`")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().yume_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
