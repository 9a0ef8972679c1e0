"""Tensor-level wrappers over the C-ABI (include/yume_hip.h).

torch is used for device memory and the current HIP stream only: every function here validates
its tensors, passes raw device pointers + sizes to libyume_hip.so and raises RuntimeError on failure.
"""
import math
import os

import torch

from . import _lib

EPI_BF16, EPI_BF16_GELU, EPI_F32, EPI_RESID, EPI_BF16_SPLITT, EPI_BF16_GELU_ERF, EPI_BF16_GEGLU = 0, 1, 2, 3, 4, 5, 6


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise RuntimeError(f"yume_amd: {name} must be a device ('cuda') tensor — this path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"yume_amd: {name} must be {dtype}, got {t.dtype}")
    return t


def _rows(t, name):
    """2-D view parameters (ptr, ld) of a tensor whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"yume_amd: {name} must be 2-D with a contiguous last dim, got shape {tuple(t.shape)} stride {t.stride()}")
    return t.data_ptr(), t.stride(0)


def _ptr(t):
    return t.data_ptr() if t is not None else None


_counters = {}
_DEBUG_COUNTERS = os.environ.get("YUME_DEBUG_COUNTERS", "0") == "1"


def ensure_counters(device):
    """Register (once per device) the caller-owned ticket-counter workspace of include/yume_hip.h (yume_counter_workspace_init): kernels
    that hand out work by ticket — long convolutions' tails, the persistent attention kernel — draw their counters from it; the library
    itself allocates nothing. The tensor lives for the life of the process.
    The zeroing is enqueued on the current stream and the DEVICE is synchronised before the workspace counts as registered: the first
    ticketed launch may come from any stream. Inside a hipGraph capture the zeroing would be recorded, not executed — refused by name
    (call yume_amd.ops.ensure_counters(device) once before capturing). YUME_DEBUG_COUNTERS=1: every call checks that the sets read zero
    (every ticketed kernel puts its zeros back; a dirty set means a launch was killed half way)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    t = _counters.get(idx)
    if t is None:
        with torch.cuda.device(idx):                       # (the capture state of THAT device's current stream)
            capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            raise RuntimeError("yume_amd: the ticket-counter workspace must be registered before a stream capture begins "
                               "(call yume_amd.ops.ensure_counters(device) once outside the capture)")
        lib = _lib.load()
        n = int(lib.yume_counter_workspace_bytes())
        with torch.cuda.device(idx):
            t = torch.zeros(n // 4, dtype=torch.int32, device=torch.device("cuda", idx))
            _lib.check(lib.yume_counter_workspace_init(t.data_ptr(), n, _stream()), "yume_counter_workspace_init")
            torch.cuda.synchronize(idx)
        _counters[idx] = t
    elif _DEBUG_COUNTERS and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize(idx)
        if int(t.abs().max()) != 0:
            raise RuntimeError("yume_amd: the ticket-counter workspace is not zero between launches (YUME_DEBUG_COUNTERS=1)")
    return t


def adaln_modulate(x, mul, add, tab_stride, row_idx, add_one, out, out_kind=0, eps=1e-6):
    """out = LN(x) * (mul[row] + add_one) + add[row]; x fp32 [T,C]; mul/add fp32 table views (first row)."""
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    xp, ldx = _rows(x, "x")
    T, C = x.shape
    op, ldo = _rows(out, "out")
    rc = lib.yume_adaln_modulate(xp, ldx, T, C, eps, mul.data_ptr(), add.data_ptr(), tab_stride, _ptr(row_idx),
                                 1 if add_one else 0, op, ldo, out_kind, _stream())
    _lib.check(rc, "yume_adaln_modulate")
    return out


# scratch of the GEMM's stream-K tail (yume_gemm_bf16_ws): one zero-initialised buffer per (device, stream) — launches sharing it are ordered
_gemm_ws = {}


def _gemm_workspace(t):
    key = _ws_key(t)
    ws = _gemm_ws.get(key)
    if ws is None:
        n = int(_lib.load().yume_gemm_workspace_bytes())
        ws = _gemm_ws[key] = torch.zeros(n, dtype=torch.uint8, device=t.device)
    return ws


def gemm_stream_k_error(device=None):
    """True if a stream-K finisher of any GEMM on `device` timed out waiting for a partial tile (the error word of the scratch)."""
    bad = False
    for (idx, _), ws in _gemm_ws.items():
        if device is None or torch.device(device).index in (None, idx):
            bad = bad or bool(ws[-64:].any().item())
    return bad


def gemm_bf16(a, w, bias, out, epi=EPI_BF16, gate=None, gate_stride=0, row_idx=None, out_t=None, n_split=0,
              variant=0):
    """acc = a @ w.T (a bf16 [M,K], w bf16 [N,K]); epilogue selected by `epi` (see yume_hip.h)."""
    lib = _lib.load()
    _dev(a, "a", torch.bfloat16)
    _dev(w, "w", torch.bfloat16)
    ap, lda = _rows(a, "a")
    wp, ldw = _rows(w, "w")
    M, K = a.shape
    N, K2 = w.shape
    if K != K2:
        raise RuntimeError(f"yume_amd.gemm_bf16: K mismatch {K} vs {K2}")
    op, ldo = _rows(out, "out")
    tp, ldt = (None, 0)
    if out_t is not None:
        tp, ldt = _rows(out_t, "out_t")
    # the scratch is handed over where the split-K plan can take the shape at all (big launches of the automatic variant). Under a stream
    # capture the first call on the capture stream allocates and zero-fills it inside the capture (the fill is replayed with the graph)
    ws = _gemm_workspace(a) if (variant == 0 and M * N >= 256 * 256 * 256) else None
    rc = lib.yume_gemm_bf16_ws(ap, lda, wp, ldw, _ptr(bias), M, N, K, epi, op, ldo, _ptr(gate), gate_stride,
                               _ptr(row_idx), tp, ldt, n_split, variant, _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    _lib.check(rc, "yume_gemm_bf16")
    return out


# Scratch buffers of the split-K GEMM and the attention key-range split are keyed by (device, stream): two callers on
# different HIP streams of one device (a VAE decode overlapped with the next chunk's denoise, say) never share partials.
_splitk_ws = {}


def _ws_key(t):
    return (t.device.index, _stream())


def gemm_small_m(a, w, bias, out, epi=EPI_BF16, target_blocks=256):
    """a @ w.T for a small number of rows against a large weight (encoders): split-K over enough slices to put ~target_blocks
    workgroups on the chip; falls back to the plain call when no split is possible or needed."""
    M, K = a.shape
    N = w.shape[0]
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    splits = 1
    while splits < 16 and tiles * splits * 2 <= target_blocks and K % (splits * 2 * 64) == 0 and K // (splits * 2) >= 256:
        splits *= 2
    if splits == 1 or N % 8 or (M * N) % 4:
        return gemm_bf16(a, w, bias, out, epi, variant=2 if epi == EPI_BF16_GEGLU else 0)
    lib = _lib.load()
    _dev(a, "a", torch.bfloat16)
    _dev(w, "w", torch.bfloat16)
    ap, lda = _rows(a, "a")
    wp, ldw = _rows(w, "w")
    op, ldo = _rows(out, "out")
    need = splits * M * N
    key = _ws_key(a)
    ws = _splitk_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 22), dtype=torch.float32, device=a.device)
        _splitk_ws[key] = ws
    rc = lib.yume_gemm_bf16_splitk(ap, lda, wp, ldw, _ptr(bias), M, N, K, epi, op, ldo, splits, ws.data_ptr(), _stream())
    _lib.check(rc, "yume_gemm_bf16_splitk")
    return out


def rmsnorm_rope(buf, C, nparts, w, eps=1e-6, rope=None, head_dim=128):
    """in-place RMSNorm over C (+RoPE) on the first nparts*C columns of bf16 `buf` [T, >=nparts*C]."""
    lib = _lib.load()
    _dev(buf, "buf", torch.bfloat16)
    bp, ld = _rows(buf, "buf")
    T = buf.shape[0]
    rc = lib.yume_rmsnorm_rope(bp, ld, T, C, nparts, w.data_ptr(), eps, _ptr(rope), head_dim, _stream())
    _lib.check(rc, "yume_rmsnorm_rope")
    return buf


def rmsnorm_rows_periodic(buf, C, w, eps=1e-6):
    """in-place RMSNorm over C of bf16 `buf` [T, C]; row t uses weight row t % w.shape[0] (w fp32 [period, C])."""
    lib = _lib.load()
    _dev(buf, "buf", torch.bfloat16)
    bp, ld = _rows(buf, "buf")
    rc = lib.yume_rmsnorm_rows_periodic(bp, ld, buf.shape[0], C, w.data_ptr(), w.shape[0], eps, _stream())
    _lib.check(rc, "yume_rmsnorm_rows_periodic")
    return buf


_attn_ws, _attn_ws_bytes = {}, {}


ATTN_Q_PRESCALED = 0x100     # YUME_ATTN_Q_PRESCALED
ATTN_KV_PADDED = 0x200       # YUME_ATTN_KV_PADDED


def attn_fwd(q, k, vt, out, Lq, Lk, H, scale=None, accumulate=False, variant=0, use_workspace=True, q_prescaled=False, kv_padded=False):
    """out[Lq, H*128] = softmax(q k^T * scale) v ; q,k token-major bf16 2-D views, vt K-major [H*128, >=Lk].
    q_prescaled: q already carries scale * log2(e) (`scale` is ignored): out = sum_j 2^(q.k_j) v_j / sum_j 2^(q.k_j).
    kv_padded: the caller guarantees that k's storage is readable up to ceil(Lk/64)*64 rows and that vt has that many columns with finite
    values behind column Lk (YUME_ATTN_KV_PADDED): with q_prescaled it opens the persistent kernel (attn_fwd8.hip)."""
    lib = _lib.load()
    _dev(q, "q", torch.bfloat16)
    _dev(k, "k", torch.bfloat16)
    _dev(vt, "vt", torch.bfloat16)
    _dev(out, "out", torch.bfloat16)
    ensure_counters(q.device)
    qp, ldq = _rows(q, "q")
    kp, ldk = _rows(k, "k")
    vp, ldv = _rows(vt, "vt")
    op, ldo = _rows(out, "out")
    if kv_padded:
        # the caller's word is checked where the tensors can tell: V^T must HAVE the padded columns, and k's storage must
        # cover the rows of the last whole tile (a tight view at the end of an allocation would be read out of bounds). That the padding
        # is finite stays the caller's contract — dit.py keeps it in zero-filled buffers.
        Lp = (Lk + 63) // 64 * 64
        if vt.shape[1] < Lp:
            raise RuntimeError(f"yume_amd.attn_fwd: kv_padded needs vt with >= {Lp} columns (ceil(Lk / 64) * 64), got {vt.shape[1]}")
        need = k.storage_offset() + (Lp - 1) * ldk + H * 128
        have = k.untyped_storage().nbytes() // k.element_size()
        if need > have:
            raise RuntimeError(f"yume_amd.attn_fwd: kv_padded needs k's storage to cover {Lp} rows (ceil(Lk / 64) * 64): the view ends "
                               f"{need - have} elements short")
    if scale is None:
        scale = 1.0 / math.sqrt(128.0)
    ws, nbytes = None, 0
    if variant in (0, 8) and use_workspace:
        key = (q.device.index, Lq, Lk, H)
        nbytes = _attn_ws_bytes.get(key)
        if nbytes is None:
            nbytes = _attn_ws_bytes[key] = int(lib.yume_attn_workspace_bytes(Lq, Lk, H))
        if nbytes:
            wkey = _ws_key(q)
            ws = _attn_ws.get(wkey)
            if ws is None or ws.numel() < nbytes:
                ws = _attn_ws[wkey] = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    rc = lib.yume_attn_fwd_ws(qp, ldq, kp, ldk, vp, ldv, op, ldo, Lq, Lk, H, scale, 1 if accumulate else 0,
                              variant | (ATTN_Q_PRESCALED if q_prescaled else 0) | (ATTN_KV_PADDED if kv_padded else 0), _ptr(ws),
                              nbytes if ws is not None else 0, _stream())
    _lib.check(rc, "yume_attn_fwd")
    return out


def linear_smallm_f32(x, w, bias, out, in_act=0, out_act=0, add_table=None):
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    R, K = x.shape
    N = w.shape[0]
    if not (x.is_contiguous() and w.is_contiguous() and out.is_contiguous()):
        raise RuntimeError("yume_amd.linear_smallm_f32: tensors must be contiguous")
    if w.dtype not in (torch.float32, torch.bfloat16) or w.shape[1] != K:
        raise RuntimeError("yume_amd.linear_smallm_f32: bad weight")
    rc = lib.yume_linear_smallm_f32(x.data_ptr(), R, K, w.data_ptr(), 1 if w.dtype == torch.bfloat16 else 0,
                                    _ptr(bias), N, in_act, out_act, _ptr(add_table), out.data_ptr(), _stream())
    _lib.check(rc, "yume_linear_smallm_f32")
    return out


def modulation_table(tab, e0, out):
    """out[b, r, :] = tab[b, :] + e0[r, :]  (fp32, contiguous)."""
    lib = _lib.load()
    _dev(tab, "tab", torch.float32)
    _dev(e0, "e0", torch.float32)
    B, W = tab.shape
    R = e0.shape[0]
    if not (tab.is_contiguous() and e0.is_contiguous() and out.is_contiguous()) or e0.shape[1] != W:
        raise RuntimeError("yume_amd.modulation_table: bad layout")
    rc = lib.yume_modulation_table(tab.data_ptr(), e0.data_ptr(), B, R, W, out.data_ptr(), _stream())
    _lib.check(rc, "yume_modulation_table")
    return out


def sinusoidal_embed(t, t_index, R, dim, out):
    lib = _lib.load()
    _dev(t, "t", torch.float64)
    rc = lib.yume_sinusoidal_embed(t.data_ptr(), _ptr(t_index), R, dim, out.data_ptr(), _stream())
    _lib.check(rc, "yume_sinusoidal_embed")
    return out


def patch_gather(x, f0, nf, kh, kw, out):
    """x [Cin,F,H,W] fp32|bf16 contiguous -> out bf16 [nf*ceil(H/kh)*ceil(W/kw), Kp]."""
    lib = _lib.load()
    _dev(x, "x")
    if not x.is_contiguous() or x.dim() != 4:
        raise RuntimeError("yume_amd.patch_gather: x must be contiguous [Cin,F,H,W]")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("yume_amd.patch_gather: x must be fp32 or bf16")
    Cin, F, H, W = x.shape
    op, Kp = _rows(out, "out")
    rc = lib.yume_patch_gather(x.data_ptr(), 1 if x.dtype == torch.bfloat16 else 0, Cin, F, H, W, f0, nf, kh, kw, op,
                               Kp, _stream())
    _lib.check(rc, "yume_patch_gather")
    return out


def unpatchify(x, Fr, Hp, Wp, ph, pw, Cout, out):
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    xp, ldi = _rows(x, "x")
    rc = lib.yume_unpatchify(xp, ldi, Fr, Hp, Wp, ph, pw, Cout, out.data_ptr(), _stream())
    _lib.check(rc, "yume_unpatchify")
    return out


def cast_bf16(x, rows_valid, out):
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    xp, ldi = _rows(x, "x")
    op, ldo = _rows(out, "out")
    rows, cols = out.shape
    rc = lib.yume_cast_bf16(xp, ldi, rows_valid, rows, cols, op, ldo, _stream())
    _lib.check(rc, "yume_cast_bf16")
    return out


def transpose_bf16(x, out):
    """x [rows, cols] fp32|bf16 -> out bf16 [cols, >=rows]."""
    lib = _lib.load()
    _dev(x, "x")
    xp, ldi = _rows(x, "x")
    op, ldo = _rows(out, "out")
    rows, cols = x.shape
    rc = lib.yume_transpose_bf16(xp, 1 if x.dtype == torch.bfloat16 else 0, ldi, rows, cols, op, ldo, _stream())
    _lib.check(rc, "yume_transpose_bf16")
    return out
