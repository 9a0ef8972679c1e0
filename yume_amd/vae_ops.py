"""Tensor-level wrappers over the VAE part of the C-ABI (include/yume_hip.h, "Causal 3D VAE" section).
Activations are bf16 channels-last tensors [T, H, W, C] (contiguous, C % 8 == 0)."""
import torch

from . import _lib
from .ops import _dev, _ptr, _stream, ensure_counters

EPI_BF16, EPI_F32, EPI_ADD, EPI_TSPLIT, EPI_RMS_SILU = 0, 2, 16, 17, 18


def _cl(t, name):
    _dev(t, name, torch.bfloat16)
    if t.dim() != 4 or not t.is_contiguous():
        raise RuntimeError(f"yume_amd.vae: {name} must be a contiguous bf16 [T,H,W,C] tensor, got {tuple(t.shape)}")
    return t


def conv3d_cl(x, cache, w, bias, cout, k, stride, pad, ups, out, epi=EPI_BF16, add=None, zero_page=None, cin=None):
    """out[To,Ho,Wo,>=cout] = conv(x [Tin,Hin,Win,C]) — geometry: k=(kt,kh,kw), stride=(st,sh,sw), pad=(pt,ph,pw)."""
    lib = _lib.load()
    _cl(x, "x")
    _cl(out, "out")
    ensure_counters(x.device)
    Tin, Hin, Win, C = x.shape
    To, Ho, Wo, Cl = out.shape
    if epi == EPI_TSPLIT:
        To = To // 2          # `out` holds the 2*To interleaved frames with cout/2 channels
    if cache is not None:
        _cl(cache, "cache")
        if tuple(cache.shape) != (2, Hin, Win, C):
            raise RuntimeError("yume_amd.vae: cache must be [2,Hin,Win,C]")
    ldadd = 0
    if epi == EPI_RMS_SILU:
        # `add` carries the fp32 gamma of the RMS_norm fused behind the convolution (include/yume_hip.h)
        _dev(add, "gamma", torch.float32)
        if add.numel() < cout or not add.is_contiguous():
            raise RuntimeError("yume_amd.vae: gamma must be a contiguous fp32 vector of >= cout elements")
    elif add is not None:
        _cl(add, "add")
        ldadd = add.shape[3]
    rc = lib.yume_conv3d_cl(x.data_ptr(), _ptr(cache), C, Tin, Hin, Win, cin if cin is not None else C, w.data_ptr(),
                            w.shape[1], _ptr(bias), cout, k[0], k[1], k[2], stride[0], stride[1], stride[2], pad[0],
                            pad[1], pad[2], 1 if ups else 0, To, Ho, Wo, epi, out.data_ptr(), Cl,
                            _ptr(add), ldadd, zero_page.data_ptr(), _stream())
    _lib.check(rc, "yume_conv3d_cl")
    return out


def rmsnorm_silu(x, gamma, silu, out, beta=None):
    lib = _lib.load()
    _cl(x, "x")
    _cl(out, "out")
    C = x.shape[3]
    M = x.numel() // C
    rc = lib.yume_vae_rmsnorm_silu(x.data_ptr(), C, M, C, gamma.data_ptr(), _ptr(beta), 1 if silu else 0, out.data_ptr(),
                                   C, _stream())
    _lib.check(rc, "yume_vae_rmsnorm_silu")
    return out


def dupup_add(x, y, ft, fs, toff):
    lib = _lib.load()
    _cl(x, "x")
    _cl(y, "y")
    Tin, Hin, Win, Cin = x.shape
    To, Ho, Wo, Cout = y.shape
    if (Ho, Wo) != (Hin * fs, Win * fs):
        raise RuntimeError("yume_amd.vae.dupup_add: spatial shape mismatch")
    rc = lib.yume_vae_dupup_add(x.data_ptr(), Cin, Tin, Hin, Win, Cin, y.data_ptr(), Cout, To, Cout, ft, fs, toff, _stream())
    _lib.check(rc, "yume_vae_dupup_add")
    return y


def avgdown_add(x, y, ft, fs):
    lib = _lib.load()
    _cl(x, "x")
    _cl(y, "y")
    Tin, Hin, Win, Cin = x.shape
    To, Ho, Wo, Cout = y.shape
    padt = (ft - Tin % ft) % ft
    if (To, Ho, Wo) != ((Tin + padt) // ft, Hin // fs, Win // fs):
        raise RuntimeError(f"yume_amd.vae.avgdown_add: shape mismatch {tuple(x.shape)} -> {tuple(y.shape)}")
    rc = lib.yume_vae_avgdown_add(x.data_ptr(), Cin, Tin, Hin, Win, Cin, y.data_ptr(), Cout, Cout, ft, fs, _stream())
    _lib.check(rc, "yume_vae_avgdown_add")
    return y


def softmax_rows(s, n, scale, p):
    lib = _lib.load()
    _dev(s, "s", torch.float32)
    _dev(p, "p", torch.bfloat16)
    rc = lib.yume_softmax_rows(s.data_ptr(), s.stride(0), s.shape[0], n, scale, p.data_ptr(), p.stride(0), _stream())
    _lib.check(rc, "yume_softmax_rows")
    return p


def pack_input(x, ps, mul, add, out):
    """x fp32|bf16 [C,T,H,W] -> out bf16 [T,H/ps,W/ps,Cpad]."""
    lib = _lib.load()
    _dev(x, "x")
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_contiguous() or x.dim() != 4:
        raise RuntimeError("yume_amd.vae.pack_input: x must be contiguous fp32|bf16 [C,T,H,W]")
    C, T, H, W = x.shape
    _cl(out, "out")
    rc = lib.yume_vae_pack_input(x.data_ptr(), 1 if x.dtype == torch.bfloat16 else 0, C, T, H, W, ps, _ptr(mul), _ptr(add),
                                 out.data_ptr(), out.shape[3], _stream())
    _lib.check(rc, "yume_vae_pack_input")
    return out


def unpack_output(x, cv, ps, sub, mul, lo, hi, out):
    """x bf16 [T,H,W,ld] (first cv channels) -> out fp32 [cv/ps^2, T, H*ps, W*ps]."""
    lib = _lib.load()
    _cl(x, "x")
    _dev(out, "out", torch.float32)
    T, H, W, ld = x.shape
    rc = lib.yume_vae_unpack_output(x.data_ptr(), ld, T, H, W, cv, ps, _ptr(sub), _ptr(mul), lo, hi, out.data_ptr(), _stream())
    _lib.check(rc, "yume_vae_unpack_output")
    return out
