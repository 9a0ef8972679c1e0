"""GPU parity of every C-ABI kernel against a plain fp32/fp64 torch restatement of the same op (computed on CPU or
with torch ops on the device in fp32). Tolerances are stated per test; bf16 outputs are compared after rounding the
reference to bf16 (|err| <= ~1 bf16 ulp of the value plus accumulation noise)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from yume_amd import ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


# ------------------------------------------------------------------------------------------- adaLN
@pytest.mark.parametrize("T,C,R", [(7, 512, 1), (300, 3072, 2), (65, 5120, 1), (33, 1280, 1)])
@pytest.mark.parametrize("out_kind", [0, 1, 2])
def test_adaln_modulate(T, C, R, out_kind):
    x = rnd(T, C, seed=1) * 3 + 0.5
    tab = rnd(R, 6, C, seed=2, scale=0.3)
    idx = (torch.arange(T) % R).to(torch.int32) if R > 1 else None
    xd, tabd = x.to(DEV), tab.to(DEV)
    idxd = idx.to(DEV) if idx is not None else None
    rows = idx.long() if idx is not None else torch.zeros(T, dtype=torch.long)
    want = torch.nn.functional.layer_norm(x.double(), (C,), eps=1e-6) * (1 + tab[rows, 1].double()) + tab[rows, 0].double()
    if out_kind == 1:
        out = torch.empty(T, C, dtype=torch.float32, device=DEV)
    elif out_kind == 0:
        out = torch.empty(T, C, dtype=torch.bfloat16, device=DEV)
    else:
        out = torch.empty(T, 3 * C, dtype=torch.bfloat16, device=DEV)
    ops.adaln_modulate(xd, tabd[:, 1], tabd[:, 0], 6 * C, idxd, True, out, out_kind)
    got = out.cpu()
    if out_kind == 1:
        assert (got.double() - want).abs().max() < 2e-5
    elif out_kind == 0:
        assert (got.double() - want).abs().max() <= 2.0 ** -8 * want.abs().max()
    else:
        hi, hi2, lo = got[:, :C], got[:, C:2 * C], got[:, 2 * C:]
        assert torch.equal(hi, hi2)
        assert ((hi.double() + lo.double()) - want).abs().max() < 3e-5 * max(1.0, want.abs().max().item())


def test_adaln_affine_mode():
    T, C = 50, 3072
    x, w, b = rnd(T, C, seed=3), 1 + 0.1 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    out = torch.empty(T, C, dtype=torch.bfloat16, device=DEV)
    ops.adaln_modulate(x.to(DEV), w.to(DEV), b.to(DEV), 0, None, False, out, 0)
    want = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-6)
    assert (out.cpu().float() - want).abs().max() <= 2.0 ** -8 * want.abs().max()


# ------------------------------------------------------------------------------------------- GEMM
def gemm_ref(a, w, bias):
    return a.double() @ w.double().t() + (bias.double() if bias is not None else 0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 192), (1, 128, 64), (129, 384, 3072), (1000, 192, 1536),
                                   (257, 512, 1280), (2048, 1024, 512), (700, 520, 4096)])
@pytest.mark.parametrize("variant", [1, 2, 3])      # 128x128, 8-wave 256x256, one-wave-per-SIMD 256x256 (gemm_w4.hpp)
def test_gemm_bf16_plain_and_f32(M, N, K, variant):
    a, w, bias = rnd(M, K, seed=1, dtype=torch.bfloat16), rnd(N, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16), rnd(N, seed=3)
    want = gemm_ref(a, w, bias)
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_bf16(ad, wd, bd, o32, ops.EPI_F32, variant=variant)
    assert rel_l2(o32.cpu(), want) < 2e-6 * math.sqrt(K) + 1e-6          # fp32 accumulation of exact bf16 products
    assert (o32.cpu().double() - want).abs().max() < 1e-4 * max(1.0, want.abs().max().item())
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_bf16(ad, wd, bd, o16, ops.EPI_BF16, variant=variant)
    assert (o16.cpu().double() - want).abs().max() <= 2.0 ** -8 * want.abs().max() + 1e-6
    og = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_bf16(ad, wd, bd, og, ops.EPI_BF16_GELU, variant=variant)
    wg = torch.nn.functional.gelu(want, approximate="tanh")
    assert (og.cpu().double() - wg).abs().max() <= 2.0 ** -7 * wg.abs().max() + 1e-5
    oe = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_bf16(ad, wd, bd, oe, ops.EPI_BF16_GELU_ERF, variant=variant)
    we = torch.nn.functional.gelu(want)
    assert (oe.cpu().double() - we).abs().max() <= 2.0 ** -7 * we.abs().max() + 1e-5


@pytest.mark.parametrize("variant", [1, 2, 3])      # 128x128, 8-wave 256x256, one-wave-per-SIMD 256x256 (gemm_w4.hpp)
def test_gemm_detects_transpose_and_permutation(variant):
    """A = I-like and asymmetric W: a swapped row/col in the C-write or a k-permutation mismatch cannot pass."""
    M, N, K = 256, 256, 256
    a = torch.eye(M, K)
    w = (torch.arange(N).view(N, 1) * 3 + torch.arange(K).view(1, K) % 7).float() / 64
    o = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_bf16(a.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV), None, o, ops.EPI_F32, variant=variant)
    assert torch.equal(o.cpu(), w.to(torch.bfloat16).float().t().contiguous())


@pytest.mark.parametrize("M,N,K,R", [(300, 512, 512, 2), (130, 3072, 1024, 1), (1100, 512, 256, 3), (1100, 512, 256, -3)])
@pytest.mark.parametrize("variant", [1, 2, 3])      # 128x128, 8-wave 256x256, one-wave-per-SIMD 256x256 (gemm_w4.hpp)
def test_gemm_resid_gate(M, N, K, R, variant):
    """R gate rows, interleaved over the tokens (R > 0) or in segments as the timesteps of a clip are (R < 0: whole waves share a row — the
    one-wave-per-SIMD kernel then loads it once — except where a segment boundary crosses a tile)."""
    segments, R = R < 0, abs(R)
    a, w, bias = rnd(M, K, seed=1, dtype=torch.bfloat16), rnd(N, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16), rnd(N, seed=3)
    x = rnd(M, N, seed=4)
    tab = rnd(R, 6, N, seed=5)
    idx = ((torch.arange(M) * R // M) if segments else (torch.arange(M) % R)).to(torch.int32)
    xd = x.to(DEV).clone()
    tabd = tab.to(DEV)
    ops.gemm_bf16(a.to(DEV), w.to(DEV), bias.to(DEV), xd, ops.EPI_RESID, gate=tabd[:, 2], gate_stride=6 * N,
                  row_idx=idx.to(DEV) if R > 1 else None, variant=variant)
    want = x.double() + gemm_ref(a, w, bias) * tab[idx.long() if R > 1 else torch.zeros(M, dtype=torch.long), 2].double()
    assert (xd.cpu().double() - want).abs().max() < 1e-4 * max(1.0, want.abs().max().item())
    xd2 = x.to(DEV).clone()
    ops.gemm_bf16(a.to(DEV), w.to(DEV), bias.to(DEV), xd2, ops.EPI_RESID, variant=variant)
    assert (xd2.cpu().double() - (x.double() + gemm_ref(a, w, bias))).abs().max() < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M", [64, 301, 516])
@pytest.mark.parametrize("variant", [1, 2, 3])      # 128x128, 8-wave 256x256, one-wave-per-SIMD 256x256 (gemm_w4.hpp)
def test_gemm_split_transposed(M, variant):
    C, K = 256, 512
    N = 3 * C
    a, w, bias = rnd(M, K, seed=1, dtype=torch.bfloat16), rnd(N, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16), rnd(N, seed=3)
    want = gemm_ref(a, w, bias)
    Mp = (M + 7) // 8 * 8
    qk = torch.empty(M, 2 * C, dtype=torch.bfloat16, device=DEV)
    vt = torch.zeros(C, Mp, dtype=torch.bfloat16, device=DEV)
    ops.gemm_bf16(a.to(DEV), w.to(DEV), bias.to(DEV), qk, ops.EPI_BF16_SPLITT, out_t=vt, n_split=2 * C, variant=variant)
    tol = 2.0 ** -8 * want.abs().max() + 1e-6
    assert (qk.cpu().double() - want[:, :2 * C]).abs().max() <= tol
    assert (vt.cpu()[:, :M].double() - want[:, 2 * C:].t()).abs().max() <= tol
    assert (vt.cpu()[:, M:] == 0).all()      # padding columns untouched


# ------------------------------------------------------------------------------------------- RMSNorm + RoPE
@pytest.mark.parametrize("T,C,nparts,rope", [(100, 512, 2, True), (37, 3072, 2, True), (64, 5120, 2, True), (50, 3072, 1, False)])
def test_rmsnorm_rope(T, C, nparts, rope):
    ld = nparts * C + 64
    buf = rnd(T, ld, seed=1, dtype=torch.bfloat16)
    w = 1 + 0.1 * rnd(nparts, C, seed=2)
    ang = rnd(T, 64, seed=3) * 3
    cs = torch.stack([torch.cos(ang.double()), torch.sin(ang.double())], dim=-1).float()
    bd = buf.to(DEV).clone()
    ops.rmsnorm_rope(bd, C, nparts, w.to(DEV), 1e-6, cs.to(DEV) if rope else None)
    got = bd.cpu()
    assert torch.equal(got[:, nparts * C:], buf[:, nparts * C:])          # columns beyond the parts untouched
    for p in range(nparts):
        x = buf[:, p * C:(p + 1) * C].double()
        y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w[p].double()
        if rope:
            yc = torch.view_as_complex(y.reshape(T, C // 128, 64, 2).contiguous())
            y = torch.view_as_real(yc * torch.polar(torch.ones_like(ang.double()), ang.double()).unsqueeze(1)).reshape(T, C)
        err = (got[:, p * C:(p + 1) * C].double() - y).abs().max()
        assert err <= 2.0 ** -8 * y.abs().max() + 1e-6, (p, err)


# ------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, scale):
    qd, kd, vd = (t.double().transpose(0, 1) for t in (q, k, v))      # [H, L, D]
    a = torch.softmax(qd @ kd.transpose(1, 2) * scale, dim=-1)
    return (a @ vd).transpose(0, 1)                                     # [Lq, H, D]


def run_attn(q, k, v, scale=None, accumulate=None, variant=0, q_prescaled=False):
    Lq, H, D = q.shape
    Lk = k.shape[0]
    vt = torch.empty(H * D, (Lk + 7) // 8 * 8, dtype=torch.bfloat16, device=DEV).fill_(float("nan"))
    ops.transpose_bf16(v.reshape(Lk, H * D).to(DEV), vt)
    out = torch.empty(Lq, H * D, dtype=torch.bfloat16, device=DEV) if accumulate is None else accumulate
    ops.attn_fwd(q.reshape(Lq, H * D).to(DEV), k.reshape(Lk, H * D).to(DEV), vt, out, Lq, Lk, H, scale=scale,
                 accumulate=accumulate is not None, variant=variant, q_prescaled=q_prescaled)
    return out.cpu().view(Lq, H, D)


@pytest.mark.parametrize("Lq,Lk,H", [(1, 1, 2), (1, 300, 2), (300, 1, 2), (5, 63, 1), (128, 64, 1), (32, 128, 2), (300, 300, 3), (1000, 512, 4), (517, 257, 2), (2048, 2048, 8),
                                     (64, 1, 1), (1, 77, 3), (130, 1000, 24), (200, 640, 2), (100, 65, 1)])
@pytest.mark.parametrize("variant", [0, 1, 2, 4, 7])
def test_attention_matches_exact_softmax(Lq, Lk, H, variant):
    q, k, v = (rnd(L, H, 128, seed=s, dtype=torch.bfloat16) for L, s in ((Lq, 1), (Lk, 2), (Lk, 3)))
    want = attn_ref(q, k, v, 1 / math.sqrt(128))
    got = run_attn(q, k, v, variant=variant)
    assert torch.isfinite(got).all()
    # P is rounded to bf16 before PV (as flash-attn does): max-abs 2^-7 of the value scale, rel-L2 well below bf16 eps
    assert (got.double() - want).abs().max() <= 1.5e-2 * max(want.abs().max().item(), 1e-3)
    assert rel_l2(got, want) < 6e-3


@pytest.mark.parametrize("variant", [0, 1, 2, 4, 7])
def test_attention_rescale_branch_and_scale(variant):
    """a key that spikes late forces the running max to jump (online-softmax rescale) in a chosen tile."""
    Lq, Lk, H = 96, 640, 2
    q, k, v = (rnd(L, H, 128, seed=s, dtype=torch.bfloat16) for L, s in ((Lq, 4), (Lk, 5), (Lk, 6)))
    k[411] = (q[17].float() * 6).to(torch.bfloat16)      # huge score for query 17 in tile 6
    k[5] = (q[40].float() * 3).to(torch.bfloat16)
    for scale in (1 / math.sqrt(128), 0.3):
        want = attn_ref(q, k, v, scale)
        got = run_attn(q, k, v, scale=scale, variant=variant)
        assert (got.double() - want).abs().max() <= 2e-2 * want.abs().max()


@pytest.mark.parametrize("variant", [0, 1, 2, 4, 7])
def test_attention_accumulate_and_transposed_operand(variant):
    Lq, Lk, H = 200, 257, 2
    q, k, v = (rnd(L, H, 128, seed=s, dtype=torch.bfloat16) for L, s in ((Lq, 7), (Lk, 8), (Lk, 9)))
    base = rnd(Lq, H * 128, seed=10, dtype=torch.bfloat16)
    got = run_attn(q, k, v, accumulate=base.to(DEV).clone(), variant=variant)
    want = attn_ref(q, k, v, 1 / math.sqrt(128)) + base.view(Lq, H, 128).double()
    assert (got.double() - want).abs().max() <= 2e-2 * want.abs().max()
    # asymmetric V (value = its own key index in d=0, head index in d=1) catches key/head permutations
    v2 = torch.zeros(Lk, H, 128)
    v2[:, :, 0] = torch.arange(Lk).view(Lk, 1) / 64.0
    v2[:, :, 1] = torch.arange(H).view(1, H) + 1.0
    v2 = v2.to(torch.bfloat16)
    got = run_attn(q, k, v2, variant=variant)
    want = attn_ref(q, k, v2, 1 / math.sqrt(128))
    assert (got.double() - want).abs().max() <= 2e-2 * want.abs().max()


@pytest.mark.parametrize("Lq,Lk,H", [(1024, 512, 2), (2100, 512, 4), (1030, 512, 24), (1500, 500, 3), (3000, 449, 2)])
@pytest.mark.parametrize("variant", [9])
def test_cross_attention_with_resident_k_and_v(Lq, Lk, H, variant):
    """attn_cross_rk_kernel (r6): the 512-key cross-attention with the head's K / V^T in registers, persistent workgroups, P^T exchanged through
    LDS in fragment order. Exact softmax against fp64; scaled and prescaled q; ragged Lq; ragged Lk in (448, 512) with the padded operands
    the engine hands over (YUME_ATTN_KV_PADDED: K readable and V^T finite up to 512); accumulate; a V that exposes any key / head permutation."""
    q, k, v = (rnd(L, H, 128, seed=s, dtype=torch.bfloat16) for L, s in ((Lq, 1), (Lk, 2), (Lk, 3)))

    def run(qq, vv, scale=None, prescaled=False, acc=None):
        kp = torch.zeros(512, H * 128, dtype=torch.bfloat16, device=DEV)
        kp[:Lk] = k.reshape(Lk, H * 128).to(DEV)
        vt = torch.zeros(H * 128, 512, dtype=torch.bfloat16, device=DEV)
        ops.transpose_bf16(vv.reshape(Lk, H * 128).to(DEV), vt)
        out = torch.empty(Lq, H * 128, dtype=torch.bfloat16, device=DEV) if acc is None else acc
        ops.attn_fwd(qq.reshape(Lq, H * 128).to(DEV), kp[:Lk], vt, out, Lq, Lk, H, scale=scale, accumulate=acc is not None, variant=variant,
                     q_prescaled=prescaled, kv_padded=True)
        return out.cpu().view(Lq, H, 128)
    want = attn_ref(q, k, v, 1 / math.sqrt(128))
    got = run(q, v)
    assert torch.isfinite(got).all()
    assert (got.double() - want).abs().max() <= 1.5e-2 * max(want.abs().max().item(), 1e-3)
    assert rel_l2(got, want) < 6e-3
    assert torch.equal(run(q, v), got)                                   # run-to-run identical
    qp = _prescale(q)
    gp = run(qp, v, prescaled=True)
    assert rel_l2(gp, attn_ref(qp, k, v, math.log(2.0))) < 6e-3
    g3 = run(q, v, scale=0.3)
    assert (g3.double() - attn_ref(q, k, v, 0.3)).abs().max() <= 2e-2 * attn_ref(q, k, v, 0.3).abs().max()
    base = rnd(Lq, H * 128, seed=10, dtype=torch.bfloat16)
    ga = run(q, v, acc=base.to(DEV).clone())
    assert (ga.double() - (want + base.view(Lq, H, 128).double())).abs().max() <= 2e-2 * (want + base.view(Lq, H, 128).double()).abs().max()
    v2 = torch.zeros(Lk, H, 128)
    v2[:, :, 0] = torch.arange(Lk).view(Lk, 1) / 64.0
    v2[:, :, 1] = torch.arange(H).view(1, H) + 1.0
    v2[:, :, 2:] = (torch.arange(126).view(1, 1, 126) % 7) * 0.25
    v2 = v2.to(torch.bfloat16)
    w2 = attn_ref(q, k, v2, 1 / math.sqrt(128))
    assert (run(q, v2).double() - w2).abs().max() <= 2e-2 * w2.abs().max()
    if variant == 9:                                                      # the same function as the streaming kernel computes
        ref2 = run_attn(q, k, v, variant=2)
        assert rel_l2(got, ref2) < 6e-3


def _prescale(q, scale=1 / math.sqrt(128)):
    """what the DiT engine hands the attention kernel: q * scale * log2(e), rounded to bf16 ONCE (there: inside yume_rmsnorm_rope)"""
    return (q.double() * (scale * math.log2(math.e))).to(torch.bfloat16)


@pytest.mark.parametrize("Lq,Lk,H", [(1, 1, 2), (300, 300, 3), (517, 257, 2), (256, 1536, 1), (300, 1600, 2), (700, 2100, 8), (512, 4096, 3), (1100, 1984, 8), (2048, 2048, 8)])
@pytest.mark.parametrize("variant", [0, 2, 7])
def test_attention_prescaled_q(Lq, Lk, H, variant):
    """YUME_ATTN_Q_PRESCALED: O = sum_j 2^(q'.k_j) v_j / sum_j 2^(q'.k_j) = softmax(ln 2 * q' k^T) v — the base-free pieces of the
    one-wave-per-SIMD kernel (variants 0 at Lk >= 1536, 7) and the same flag through the other kernels (scale_log2 = 1)."""
    q, k, v = (rnd(L, H, 128, seed=s, dtype=torch.bfloat16) for L, s in ((Lq, 1), (Lk, 2), (Lk, 3)))
    qp = _prescale(q)
    want = attn_ref(qp, k, v, math.log(2.0))
    got = run_attn(qp, k, v, variant=variant, q_prescaled=True)
    assert torch.isfinite(got).all()
    assert (got.double() - want).abs().max() <= 1.5e-2 * max(want.abs().max().item(), 1e-3)
    assert rel_l2(got, want) < 6e-3
    # and it is the function the unscaled call computes, up to the one extra rounding of q this test (not the engine) pays
    plain = run_attn(q, k, v, variant=variant)
    assert rel_l2(got, plain) < 1.2e-2


@pytest.mark.parametrize("variant", [0, 7])
@pytest.mark.parametrize("Lq,Lk,H", [(300, 1600, 2), (700, 2100, 8), (256, 64, 1), (8500, 2100, 8)])
def test_attention_prescaled_q_out_of_range_rows_take_the_robust_pieces(Lq, Lk, H, variant):
    """scores far outside what exp2 can hold without a base: one query with a score of several hundred (its exponential is inf), one whose
    scores all sit near -400 (every exponential flushes to 0), one at +100 (fine without a base). The workgroups holding the first two
    fail the range check at their end and are rerun, inside the launch, on the rescaling pieces; all rows match the exact softmax."""
    q, k, v = (rnd(L, H, 128, seed=s, dtype=torch.bfloat16) for L, s in ((Lq, 11), (Lk, 12), (Lk, 13)))
    q, k = q.float(), k.float()
    k[:, :, 0] = 8.0
    q[Lq - 2, :, 0] = -400.0
    k[5] = q[3] * 40
    k[Lk - 9] = q[Lq // 2] * 6
    q, k = q.to(torch.bfloat16), k.to(torch.bfloat16)
    qp = _prescale(q)
    want = attn_ref(qp, k, v, math.log(2.0))
    got = run_attn(qp, k, v, variant=variant, q_prescaled=True)
    assert torch.isfinite(got).all()
    assert (got.double() - want).abs().max() <= 2e-2 * want.abs().max()
    assert rel_l2(got, want) < 6e-3


def test_attention_auto_splits_query_range_between_kernels():
    """variant 0 at Lk >= 1536: 256-query workgroups of the one-wave-per-SIMD kernel; the query blocks of a partial last round
    are cut into key ranges inside the same launch (here 1 head per XCD x 34 blocks = 1 round of 32 + 2 -> rows [0, 8192) whole,
    [8192, 8500) as two key halves each)."""
    Lq, Lk, H = 8500, 2100, 8
    q, k, v = rnd(Lq, H * 128, seed=1, dtype=torch.bfloat16), rnd(Lk, H * 128, seed=2, dtype=torch.bfloat16), rnd(Lk, H * 128, seed=3, dtype=torch.bfloat16)
    base = rnd(Lq, H * 128, seed=4, dtype=torch.bfloat16).to(DEV)
    for acc in (False, True):
        want = run_attn(q.view(Lq, H, 128), k.view(Lk, H, 128), v.view(Lk, H, 128), accumulate=base.clone() if acc else None, variant=2)
        got = run_attn(q.view(Lq, H, 128), k.view(Lk, H, 128), v.view(Lk, H, 128), accumulate=base.clone() if acc else None, variant=0)
        assert rel_l2(got.float(), want.float()) < 2e-3
        assert (got.float() - want.float()).abs().max() <= 2.0 ** -6 * want.float().abs().max()
    # the tail rows went through the key-range split (two partial softmaxes merged): same function as the unsplit call
    qd, kd = q.to(DEV), k.to(DEV)
    vt = torch.zeros(H * 128, (Lk + 7) // 8 * 8, dtype=torch.bfloat16, device=DEV)
    ops.transpose_bf16(v.to(DEV), vt)
    o_ws = torch.empty(Lq, H * 128, dtype=torch.bfloat16, device=DEV)
    o_no = torch.empty_like(o_ws)
    ops.attn_fwd(qd, kd, vt, o_ws, Lq, Lk, H, use_workspace=True)
    ops.attn_fwd(qd, kd, vt, o_no, Lq, Lk, H, use_workspace=False)
    assert torch.equal(o_ws[:8192], o_no[:8192])
    assert rel_l2(o_ws[8192:].float().cpu(), o_no[8192:].float().cpu()) < 4e-3      # two bf16 roundings of fp32 values that differ in the last bits


# ---- the persistent kernel (attn_fwd8.hip): variant 8, YUME_ATTN_Q_PRESCALED | YUME_ATTN_KV_PADDED -----------------------------
def run_attn_padded(q, k, v, accumulate=None, variant=8, use_workspace=True, k_pad=float("nan"), v_pad=37.5):
    """q [Lq, H, 128] PRESCALED bf16; K handed over with rows up to a whole 64-key tile (k_pad: NaN — nobody may use them), V^T with that many
    columns holding FINITE junk behind column Lk (the contract of YUME_ATTN_KV_PADDED)."""
    Lq, H, D = q.shape
    Lk = k.shape[0]
    Lp = (Lk + 63) // 64 * 64
    kp = torch.full((Lp, H * D), k_pad, dtype=torch.bfloat16, device=DEV)
    kp[:Lk] = k.reshape(Lk, H * D).to(DEV)
    vt = torch.full((H * D, Lp), v_pad, dtype=torch.bfloat16, device=DEV)
    ops.transpose_bf16(v.reshape(Lk, H * D).to(DEV), vt)
    out = torch.empty(Lq, H * D, dtype=torch.bfloat16, device=DEV) if accumulate is None else accumulate
    ops.attn_fwd(q.reshape(Lq, H * D).to(DEV), kp[:Lk], vt, out, Lq, Lk, H, accumulate=accumulate is not None, variant=variant, q_prescaled=True,
                 kv_padded=True, use_workspace=use_workspace)
    return out.cpu().view(Lq, H, D)


# shapes: whole tiles / ragged last tile; 8-tile items (the cross-attention length) and long ones; fewer heads than XCDs (idle queues are
# stolen from), more than one head per XCD, a partial last query block, more items than CUs (streaming across items), exactly 4..7 tiles
# in the last piece
V8_SHAPES = [(256, 512, 1), (300, 515, 2), (1000, 640, 3), (2048, 2048, 8), (700, 2100, 8), (1100, 1984, 9), (4000, 2500, 12), (9460, 512, 24), (5000, 1000, 24),
             (70000, 576, 1)]


@pytest.mark.parametrize("Lq,Lk,H", V8_SHAPES)
def test_attention_persistent_kernel_matches_exact_softmax_and_variant_7(Lq, Lk, H):
    q, k, v = (rnd(L, H, 128, seed=s, dtype=torch.bfloat16) for L, s in ((Lq, 21), (Lk, 22), (Lk, 23)))
    qp = _prescale(q)
    got = run_attn_padded(qp, k, v)
    assert torch.isfinite(got).all()
    if Lq * Lk * H <= 4000 * 2500 * 12:
        want = attn_ref(qp, k, v, math.log(2.0))
        assert (got.double() - want).abs().max() <= 1.5e-2 * max(want.abs().max().item(), 1e-3)
        assert rel_l2(got, want) < 6e-3
    else:
        ref = run_attn(qp, k, v, variant=2, q_prescaled=True)
        assert rel_l2(got.float(), ref.float()) < 3e-3
    # the same arithmetic per key tile in the same order as attn_fwd7: without scratch (whole query blocks only) the two agree bit for bit
    a8 = run_attn_padded(qp, k, v, use_workspace=False)
    a7 = run_attn(qp, k, v, variant=7, q_prescaled=True)
    assert torch.equal(a8, a7)
    # variant 0 with both flags takes the same kernel where attn_fwd7 was its choice (Lk >= 1536; shorter key sequences — the 512-key
    # cross-attention — stay on the 4-wave kernel, which measured faster there)
    auto = run_attn_padded(qp, k, v, variant=0)
    if Lk >= 1536:
        assert torch.equal(auto, got)
    else:
        assert rel_l2(auto.float(), got.float()) < 6e-3         # two kernels, two bf16 roundings of P and of O


def test_attention_persistent_kernel_accumulate_and_repeated_launches():
    """accumulate (the 14B image cross-attention sum) and the counter workspace's invariant: 300 launches in a row (more than the 256 counter
    sets: every set is reused) keep giving the same bits — a launch that left its tickets behind would starve or repeat items."""
    Lq, Lk, H = 3000, 1100, 5
    q, k, v = (rnd(L, H, 128, seed=s, dtype=torch.bfloat16) for L, s in ((Lq, 31), (Lk, 32), (Lk, 33)))
    qp = _prescale(q)
    base = rnd(Lq, H * 128, seed=34, dtype=torch.bfloat16)
    got = run_attn_padded(qp, k, v, accumulate=base.to(DEV).clone())
    want = attn_ref(qp, k, v, math.log(2.0)) + base.view(Lq, H, 128).double()
    assert (got.double() - want).abs().max() <= 2e-2 * want.abs().max()
    first = run_attn_padded(qp, k, v)
    for _ in range(300):
        again = run_attn_padded(qp, k, v)
    assert torch.equal(again, first)
    from yume_amd.ops import _counters
    torch.cuda.synchronize()
    assert int(_counters[torch.cuda.current_device()].abs().sum()) == 0          # every launch put its zeros back


@pytest.mark.parametrize("Lq,Lk,H", [(700, 2100, 8), (4000, 1600, 3), (8500, 2100, 8)])
def test_attention_persistent_kernel_out_of_range_items_rerun_on_the_robust_pieces(Lq, Lk, H):
    """as test_attention_prescaled_q_out_of_range_rows_take_the_robust_pieces: the items whose range vote fails are redone cold on attn_fwd7's
    rescaling pieces inside the launch and the stream restarts behind them."""
    q, k, v = (rnd(L, H, 128, seed=s, dtype=torch.bfloat16) for L, s in ((Lq, 11), (Lk, 12), (Lk, 13)))
    q, k = q.float(), k.float()
    k[:, :, 0] = 8.0
    q[Lq - 2, :, 0] = -400.0
    k[5] = q[3] * 40
    k[Lk - 9] = q[Lq // 2] * 6
    q, k = q.to(torch.bfloat16), k.to(torch.bfloat16)
    qp = _prescale(q)
    want = attn_ref(qp, k, v, math.log(2.0))
    got = run_attn_padded(qp, k, v)
    assert torch.isfinite(got).all()
    assert (got.double() - want).abs().max() <= 2e-2 * want.abs().max()
    assert rel_l2(got, want) < 6e-3
    # determinism of the rerun path: the robust rerun inside the persistent launch shares LDS and the DMA queue with the stream
    # it interrupts; twenty more launches on the same data have to give the same bits as the first, in the rerun items and around them
    for _ in range(20):
        assert torch.equal(run_attn_padded(qp, k, v), got)
    # ... and the fallback kernel (attn_fwd7: same rerun pieces, no stream) is deterministic on the same data as well
    v7 = run_attn(qp, k, v, variant=7, q_prescaled=True)
    for _ in range(5):
        assert torch.equal(run_attn(qp, k, v, variant=7, q_prescaled=True), v7)


def test_attention_persistent_kernel_refuses_what_it_cannot_take():
    q, k, v = (rnd(L, 2, 128, seed=s, dtype=torch.bfloat16) for L, s in ((300, 1), (300, 2), (300, 3)))
    with pytest.raises(RuntimeError, match="variant 8"):
        run_attn_padded(_prescale(q), k, v)                       # Lk < 512


def test_flash_attention_seam():
    from yume_amd.attention import flash_attention
    B, Lq, Lk, H = 2, 150, 90, 3
    q, k, v = rnd(B, Lq, H, 128, seed=1), rnd(B, Lk, H, 128, seed=2), rnd(B, Lk, H, 128, seed=3)
    out = flash_attention(q.to(DEV), k.to(DEV), v.to(DEV), k_lens=torch.tensor([90, 61]))
    assert out.dtype == torch.float32 and out.shape == (B, Lq, H, 128)
    for b, lk in ((0, 90), (1, 61)):
        want = attn_ref(q[b].bfloat16(), k[b, :lk].bfloat16(), v[b, :lk].bfloat16(), 1 / math.sqrt(128))
        assert (out[b].cpu().double() - want).abs().max() <= 2e-2 * want.abs().max()


# ------------------------------------------------------------------------------------------- small helpers
def test_sinusoidal_and_time_mlp():
    t = torch.tensor([0.0, 731.4285714285714, 1000.0, 3.25], dtype=torch.float64)
    idx = torch.tensor([3, 0, 1], dtype=torch.int32)
    out = torch.empty(3, 256, dtype=torch.float32, device=DEV)
    ops.sinusoidal_embed(t.to(DEV), idx.to(DEV), 3, 256, out)
    half = 128
    w = torch.pow(10000.0, -torch.arange(half, dtype=torch.float64) / half)
    a = torch.outer(t[idx.long()], w)
    want = torch.cat([torch.cos(a), torch.sin(a)], dim=1).float()
    assert (out.cpu() - want).abs().max() < 1e-6
    for wd in (torch.float32, torch.bfloat16):
        R, K, N = 3, 256, 520
        x, W, b, add = rnd(R, K, seed=1), rnd(N, K, seed=2, scale=0.05).to(wd), rnd(N, seed=3), rnd(N, seed=4)
        o = torch.empty(R, N, dtype=torch.float32, device=DEV)
        ops.linear_smallm_f32(x.to(DEV), W.to(DEV), b.to(DEV), o, in_act=1, out_act=1, add_table=add.to(DEV))
        want = torch.nn.functional.silu(torch.nn.functional.silu(x.double()) @ W.double().t() + b.double()) + add.double()
        assert (o.cpu().double() - want).abs().max() < 1e-5


def test_modulation_table():
    tab, e0 = rnd(5, 48, seed=1), rnd(3, 48, seed=2)
    out = torch.empty(5, 3, 48, device=DEV)
    ops.modulation_table(tab.to(DEV), e0.to(DEV), out)
    assert torch.equal(out.cpu(), tab[:, None, :] + e0[None, :, :])


@pytest.mark.parametrize("k,dtype", [(2, torch.float32), (4, torch.float32), (8, torch.bfloat16), (32, torch.float32)])
def test_patch_gather_matches_conv3d(k, dtype):
    Cin, F, H, W, Co = 6, 5, 11, 14, 8
    x = rnd(Cin, F, H, W, seed=1).to(dtype)
    wt = rnd(Co, Cin, 1, k, k, seed=2)
    f0, nf = 1, 3
    Hp, Wp = -(-H // k), -(-W // k)
    K = Cin * k * k
    Kp = (K + 63) // 64 * 64
    out = torch.empty(nf * Hp * Wp, Kp, dtype=torch.bfloat16, device=DEV)
    ops.patch_gather(x.to(DEV), f0, nf, k, k, out)
    xp = torch.nn.functional.pad(x[:, f0:f0 + nf].float(), (0, Wp * k - W, 0, Hp * k - H)).to(torch.bfloat16).float()
    want = torch.nn.functional.conv3d(xp.unsqueeze(0), wt, stride=(1, k, k))[0].flatten(1).t()
    got = out.cpu().float()[:, :K] @ wt.flatten(1).t()
    assert (got - want).abs().max() < 1e-3
    assert (out.cpu()[:, K:] == 0).all()


def test_unpatchify_cast_transpose():
    Fr, Hp, Wp, Co = 3, 4, 5, 6
    y = rnd(Fr * Hp * Wp, 4 * Co, seed=1)
    out = torch.empty(Co, Fr, 2 * Hp, 2 * Wp, device=DEV)
    ops.unpatchify(y.to(DEV), Fr, Hp, Wp, 2, 2, Co, out)
    want = torch.einsum("fhwpqrc->cfphqwr", y.view(Fr, Hp, Wp, 1, 2, 2, Co)).reshape(Co, Fr, 2 * Hp, 2 * Wp)
    assert torch.equal(out.cpu(), want)
    x = rnd(5, 64, seed=2)
    o = torch.empty(9, 64, dtype=torch.bfloat16, device=DEV)
    ops.cast_bf16(x.to(DEV), 5, o)
    assert torch.equal(o.cpu()[:5], x.to(torch.bfloat16)) and (o.cpu()[5:] == 0).all()
    for dt in (torch.float32, torch.bfloat16):
        z = rnd(70, 45, seed=3).to(dt)
        ot = torch.zeros(45, 72, dtype=torch.bfloat16, device=DEV)
        ops.transpose_bf16(z.to(DEV), ot)
        assert torch.equal(ot.cpu()[:, :70], z.to(torch.bfloat16).t())


def test_errors_are_loud():
    a = torch.zeros(4, 60, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(8, 60, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(4, 8, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiple"):
        ops.gemm_bf16(a, w, None, o)
    with pytest.raises(RuntimeError, match="device"):
        ops.gemm_bf16(a.cpu(), w, None, o)


def test_gemm_auto_row_split_between_kernels():
    """variant 0 when the 256x256 tiling leaves a nearly empty last round (here 37 x 28 = 1036 tiles = 4 rounds + 12): whole
    rounds of M-tiles on the 256x256 kernel, the remaining rows on the 128x128 kernel — every epilogue's row-dependent operand
    (residual rows, gate row selector, K-major transposed columns) must follow the split."""
    M, N, K = 9460, 7168, 256
    a, w, bias = rnd(M, K, seed=1, dtype=torch.bfloat16).to(DEV), rnd(N, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16).to(DEV), rnd(N, seed=3).to(DEV)
    x = rnd(M, N, seed=4).to(DEV)
    tab = rnd(2, 6, N, seed=5).to(DEV)
    idx = (torch.arange(M) % 2).to(torch.int32).to(DEV)
    for variant_ref in (2,):
        want = ops.gemm_bf16(a, w, bias, x.clone(), ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * N, row_idx=idx, variant=variant_ref)
        got = ops.gemm_bf16(a, w, bias, x.clone(), ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * N, row_idx=idx, variant=0)
        assert (got - want).abs().max() <= 2e-5 * want.abs().max()
    Mp = (M + 7) // 8 * 8
    ns = N - 1024

    def splitt(variant):
        qk = torch.empty(M, ns, dtype=torch.bfloat16, device=DEV)
        vt = torch.zeros(N - ns, Mp, dtype=torch.bfloat16, device=DEV)
        ops.gemm_bf16(a, w, bias, qk, ops.EPI_BF16_SPLITT, out_t=vt, n_split=ns, variant=variant)
        return qk.float(), vt.float()
    (q2, v2), (q0, v0) = splitt(2), splitt(0)
    assert (q0 - q2).abs().max() <= 2.0 ** -7 * q2.abs().max() and (v0 - v2).abs().max() <= 2.0 ** -7 * v2.abs().max()
    assert (v0[:, M:] == 0).all()
    og, og2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_bf16(a, w, bias, og, ops.EPI_BF16_GELU, variant=0)
    ops.gemm_bf16(a, w, bias, og2, ops.EPI_BF16_GELU, variant=2)
    assert (og.float() - og2.float()).abs().max() <= 2.0 ** -7 * og2.float().abs().max()


@pytest.mark.parametrize("M,N,K", [(77, 4096, 4096), (512, 1280, 5120), (257, 2048, 1024), (3, 256, 512)])
def test_gemm_small_m_split_k_matches_plain(M, N, K):
    """split-K path of the encoders' small-M GEMMs: every supported epilogue against the one-pass kernels."""
    a = rnd(M, K, seed=1, dtype=torch.bfloat16).to(DEV)
    w = rnd(N, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16).to(DEV)
    bias = rnd(N, seed=3).to(DEV)
    for epi, odt, ncol in ((ops.EPI_F32, torch.float32, N), (ops.EPI_BF16, torch.bfloat16, N), (ops.EPI_BF16_GELU_ERF, torch.bfloat16, N),
                           (ops.EPI_BF16_GELU, torch.bfloat16, N)):
        want = ops.gemm_bf16(a, w, bias, torch.empty(M, ncol, dtype=odt, device=DEV), epi)
        got = ops.gemm_small_m(a, w, bias, torch.empty(M, ncol, dtype=odt, device=DEV), epi)
        tol = 2e-5 if odt == torch.float32 else 2.0 ** -7
        assert (got.float() - want.float()).abs().max() <= tol * max(1.0, want.float().abs().max().item())
    x = rnd(M, N, seed=4).to(DEV)
    want = ops.gemm_bf16(a, w, bias, x.clone(), ops.EPI_RESID)
    got = ops.gemm_small_m(a, w, bias, x.clone(), ops.EPI_RESID)
    assert (got - want).abs().max() <= 2e-5 * want.abs().max()
    want = ops.gemm_bf16(a, w, None, torch.empty(M, N // 2, dtype=torch.bfloat16, device=DEV), ops.EPI_BF16_GEGLU, variant=2)
    got = ops.gemm_small_m(a, w, None, torch.empty(M, N // 2, dtype=torch.bfloat16, device=DEV), ops.EPI_BF16_GEGLU)
    assert (got.float() - want.float()).abs().max() <= 2.0 ** -7 * max(1.0, want.float().abs().max().item())
    # and against the exact product
    ref = gemm_ref(a.cpu(), w.cpu(), bias.cpu())
    o32 = ops.gemm_small_m(a, w, bias, torch.empty(M, N, dtype=torch.float32, device=DEV), ops.EPI_F32)
    assert (o32.cpu().double() - ref).abs().max() < 1e-4 * max(1.0, ref.abs().max().item())


def test_calibration_microkernel_reports_a_plausible_sustained_rate():
    """yume_calibrate_mfma through yume_amd.calibrate (bench.py's `calibration`): the pure-MFMA rate on random operands must lie between a
    tenth of and a little above the nominal 2.5 PFLOP/s and the fixed 8192^3 reference launch must land where the product GEMM does. The
    kernel's own s_memtime reading is printed next to the implied clock (informational: whether s_memtime ticks at the shader clock is a
    firmware property, not the product's — ADVICE r5)."""
    from yume_amd import calibrate
    m = calibrate.mfma_sustained(DEV, settle_s=0.05, measure_s=0.1)
    assert 250.0 < m["tflops"] < 2700.0 and 0.25 < m["clock_ghz"] < 2.6
    print(f"calibration: {m['tflops']:.0f} TF/s, implied clock {m['clock_ghz']:.3f} GHz, s_memtime {m['s_memtime_ghz']} GHz, measured {m['measured_s']:.3f} s")
    assert m["measured_s"] >= 0.08
    g = calibrate.gemm_reference(DEV, reps=5)
    assert 300.0 < g["tflops"] < 2500.0
