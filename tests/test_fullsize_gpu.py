"""Full-size (BASELINE configs[1] = Yume-5B-720P, L = 9460, 30 blocks) checks that do not need the slow CPU oracle on the
whole model: size-independent properties of the kernels at the real shapes, the complete 30-block engine with the blocks
made exact identities (so the oracle only has to evaluate the embeddings and the head), run-to-run reproducibility, and
the causal property of the full-resolution Wan2.2 decoder."""
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)

from oracle import dit as odit  # noqa: E402
from yume_amd import ops, synth  # noqa: E402

DEV = "cuda"
L5B, C5B, H5B, FF5B = 9460, 3072, 24, 14336


def bf(*shape, seed=0, scale=0.5):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * scale).to(torch.bfloat16)


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


# ------------------------------------------------------------------------------------------ attention, 24 heads x 9460
def _attn(q, k, vt, Lq, Lk, H, variant=0, use_workspace=True):
    o = torch.empty(Lq, H * 128, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd(q, k, vt, o, Lq, Lk, H, variant=variant, use_workspace=use_workspace)
    return o


def test_attention_full_size_properties():
    L, H, C = L5B, H5B, C5B
    Lp = (L + 7) // 8 * 8
    q, k = bf(L, C, seed=1), bf(L, C, seed=2)
    # (a) softmax rows sum to one: a V that is constant over the keys comes back unchanged (to bf16 rounding of P)
    const = bf(C, 1, seed=3, scale=1.0)
    vt = const.expand(C, Lp).contiguous()
    o = _attn(q, k, vt, L, L, H)
    want = const.float().view(1, C).expand(L, C)
    assert (o.float() - want).abs().max() <= 2.0 ** -7 * want.abs().max()
    # (b) a permutation of the keys (rows of K, columns of V^T) does not change the result beyond summation order
    vt = torch.zeros(C, Lp, dtype=torch.bfloat16, device=DEV)
    vt[:, :L] = bf(C, L, seed=4)
    o1 = _attn(q, k, vt, L, L, H)
    perm = torch.randperm(L, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    vt2 = torch.zeros_like(vt)
    vt2[:, :L] = vt[:, :L][:, perm]
    o2 = _attn(q, k[perm].contiguous(), vt2, L, L, H)
    assert rel_l2(o2, o1) < 4e-3
    # (c) a query's result does not depend on how many other queries are in the launch (rows 4000.. sit in whole query blocks of
    # the full launch; without scratch the small launch computes whole blocks too — with it, its blocks would be cut into key
    # ranges, which is the same function with another summation order, checked to rounding)
    o3 = _attn(q[4000:4300].contiguous(), k, vt, 300, L, H, use_workspace=False)
    assert torch.equal(o3, o1[4000:4300])
    o3s = _attn(q[4000:4300].contiguous(), k, vt, 300, L, H)
    assert rel_l2(o3s, o3) < 4e-3
    # (d) the register-staged kernel (variant 1) is an independent implementation of the same arithmetic
    o4 = _attn(q, k, vt, L, L, H, variant=1)
    assert rel_l2(o4, o1) < 2e-3
    # (e) keys >= Lk are ignored whatever they hold
    k_bad, vt_bad = k.clone(), vt.clone()
    k_bad[9000:] = 1e4
    vt_bad[:, 9000:] = float("nan")
    o5 = _attn(q, k_bad, vt_bad, L, 9000, H)
    o6 = _attn(q, k[:9000].contiguous(), vt[:, :9000].contiguous(), L, 9000, H)
    assert torch.isfinite(o5.float()).all() and torch.equal(o5, o6)


# ------------------------------------------------------------------------------------------ GEMMs of the 5B block
@pytest.mark.parametrize("name,M,N,K", [("qkv", L5B, 3 * C5B, C5B), ("ffn0", L5B, FF5B, C5B), ("ffn2", L5B, C5B, FF5B)])
def test_gemm_full_size_properties(name, M, N, K):
    a, w = bf(M, K, seed=1), bf(N, K, seed=2, scale=K ** -0.5)
    bias = torch.randn(N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    o256 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_bf16(a, w, bias, o256, ops.EPI_F32, variant=2)
    # (a) the 128x128 kernel is a second, independent tiling of the same sum (same K order inside a tile row)
    o128 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_bf16(a, w, bias, o128, ops.EPI_F32, variant=1)
    assert rel_l2(o128, o256) < 1e-6
    # (b) a row's result does not depend on M (ragged last tile included): same kernel, 301 rows instead of 9460
    sub = torch.empty(301, N, dtype=torch.float32, device=DEV)
    ops.gemm_bf16(a[M - 301:].contiguous(), w, bias, sub, ops.EPI_F32, variant=2)
    assert torch.equal(sub, o256[M - 301:])
    # (c) exact reference on a sample of rows (fp64 on the host)
    rows = torch.tensor([0, 1, 255, 256, 4097, M - 245, M - 1])
    want = a[rows.to(DEV)].double().cpu() @ w.double().cpu().t() + bias.double().cpu()
    assert (o256[rows.to(DEV)].double().cpu() - want).abs().max() <= 1e-4 * max(1.0, want.abs().max().item())
    # (d) scaling A by two (exact in bf16 and in every fp32 partial sum) doubles the result bit for bit
    o1 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    o2 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_bf16(a, w, None, o1, ops.EPI_F32, variant=2)
    ops.gemm_bf16((a.float() * 2).to(torch.bfloat16), w, None, o2, ops.EPI_F32, variant=2)
    assert torch.equal(o2, 2 * o1)


# ------------------------------------------------------------------------------------------ the 30-block engine
def _identity_blocks_(model):
    """Make every block an exact identity on the residual stream while all of its kernels still run at full size:
    gates = modulation rows 2, 5 + time projection -> 0; the un-gated cross-attention is silenced through its o-proj."""
    with torch.no_grad():
        model.time_projection[1].weight.zero_()
        model.time_projection[1].bias.zero_()
        for b in model.blocks:
            b.modulation[:, 2].zero_()
            b.modulation[:, 5].zero_()
            b.cross_attn.o.weight.zero_()
            b.cross_attn.o.bias.zero_()


def test_full_5b_engine_identity_blocks_vs_oracle_and_reproducible():
    from yume_amd import framepack
    from yume_amd.wan23.modules.model import WanModel
    cfg = dict(synth.CFG_5B)
    with torch.device(DEV):
        model = WanModel(**cfg)
    synth.randomize_module_(model, seed=3)
    _identity_blocks_(model)
    model = model.eval().requires_grad_(False)
    F, H, W, lfz = 13, 44, 80, 8
    plan = framepack.pack_plan(F, H, W, lfz)
    assert plan.seq_len == L5B
    g = torch.Generator().manual_seed(9)
    x = torch.randn(48, F, H, W, generator=g)
    ctx = torch.randn(77, 4096, generator=g)
    t = torch.cat([torch.zeros(plan.n_hist_tok), torch.full((plan.n_new_tok,), 731.0)]).unsqueeze(0)

    def run():
        return model([x.to(DEV)], t=t.to(DEV), context=[ctx.to(DEV)], seq_len=L5B, latent_frame_zero=lfz, flag=True)[0]
    got = run()
    assert got.shape == (48, lfz, H, W) and torch.isfinite(got).all()
    assert torch.equal(run(), got)                               # same launch sequence -> same bits
    # oracle: identity blocks == no blocks
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if not k.startswith("blocks.")}
    cfg0 = dict(cfg, num_layers=0)
    want = odit.forward_wan23(sd, cfg0, x, t, ctx, L5B, lfz, True)
    e = rel_l2(got.cpu(), want)
    print(f"full 5B-c0, identity blocks: rel-L2 {e:.3e}")
    assert e <= 5e-3
    # and with live blocks the same engine must still be reproducible and finite
    synth.randomize_module_(model, seed=4)
    a, b = run(), run()
    assert torch.isfinite(a).all() and torch.equal(a, b)


# ------------------------------------------------------------------------------------------ Wan2.2 decoder, 704 x 1280
def test_full_resolution_decoder_is_causal():
    """Frames decoded from the first k latents do not change when later latents are appended (CausalConv3d + the
    2-frame feature cache, vae2_2.py:831-860): decode(z[:, :3]) == decode(z[:, :8])[:, :9] at 704x1280."""
    from yume_amd.wan23.modules.vae2_2 import Wan2_2_VAE, WanVAE_
    cfg = synth.VAE_CFG_22
    with torch.device(DEV):
        m = WanVAE_(dim=cfg["dim"], dec_dim=cfg["dec_dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
    g = torch.Generator(device=DEV).manual_seed(0)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("gamma"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device=DEV))
            elif k.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g, device=DEV))
            else:
                p.copy_((torch.rand(p.shape, generator=g, device=DEV) * 2 - 1) * (3.0 / p[0].numel()) ** 0.5)
    vae = Wan2_2_VAE(device=DEV, model=m)
    z = torch.randn(48, 8, 44, 80, device=DEV, generator=g)
    full = vae.decode([z])[0]
    head = vae.decode([z[:, :3].contiguous()])[0]
    assert full.shape == (3, 29, 704, 1280) and head.shape == (3, 9, 704, 1280)
    assert torch.isfinite(full).all() and full.abs().max() <= 1.0
    assert torch.equal(head, full[:, :9])


def test_full_14b_engine_identity_blocks_vs_oracle():
    """BASELINE configs[2] geometry (Yume-I2V-14B-540P, latent [16,17,68,120] + y, FramePack lfz=9, L = 27810, 40 blocks
    of width 5120) with identity blocks: embeddings (x|y concat, pyramid, CLIP image tokens) + head against the oracle."""
    from yume_amd import framepack
    from yume_amd.wan.modules.model import WanModel
    cfg = dict(synth.CFG_14B)
    with torch.device(DEV):
        model = WanModel(**cfg).attach_pyramid()
    synth.randomize_module_(model, seed=5)
    _identity_blocks_(model)
    model = model.eval().requires_grad_(False)
    F, H, W, lfz = 17, 68, 120, 9
    L = framepack.pack_plan(F, H, W, lfz, F - 9).seq_len
    assert L == 27810
    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(16, F, H, W, generator=g), torch.randn(20, F, H, W, generator=g)
    ctx, clip = torch.randn(77, 4096, generator=g), torch.randn(1, 257, 1280, generator=g)
    t = torch.tensor([612.0])
    got, cache = model([x.to(DEV)], t=t.to(DEV), context=[ctx.to(DEV)], seq_len=L, clip_fea=clip.to(DEV), y=[y.to(DEV)],
                       rand_num_img=0.6, latent_frame_zero=lfz)
    assert cache is None and got.shape == (16, lfz, H, W) and torch.isfinite(got).all()
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if not k.startswith("blocks.")}
    want = odit.forward_wan(sd, dict(cfg, num_layers=0), x, t, ctx, L, clip[0], y, 0.6, lfz)
    e = rel_l2(got.cpu(), want)
    print(f"full 14B (configs[2] geometry), identity blocks: rel-L2 {e:.3e}")
    assert e <= 5e-3


def test_full_resolution_wan21_decoder_is_causal():
    """Wan2.1 VAE at 544x960 (the 14B pipeline's decoder, vae.py:544-568): decode(z[:, :4]) == decode(z[:, :13])[:, :13]."""
    from yume_amd.wan.modules.vae import WanVAE, WanVAE_
    cfg = synth.VAE_CFG_21
    with torch.device(DEV):
        m = WanVAE_(dim=cfg["dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
    g = torch.Generator(device=DEV).manual_seed(1)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("gamma"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device=DEV))
            elif k.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g, device=DEV))
            else:
                p.copy_((torch.rand(p.shape, generator=g, device=DEV) * 2 - 1) * (3.0 / p[0].numel()) ** 0.5)
    vae = WanVAE(z_dim=cfg["z_dim"], device=DEV, model=m)
    z = torch.randn(16, 13, 68, 120, device=DEV, generator=g)
    full = vae.decode([z])[0]
    head = vae.decode([z[:, :4].contiguous()])[0]
    assert full.shape == (3, 49, 544, 960) and head.shape == (3, 13, 544, 960)
    assert torch.isfinite(full).all() and full.abs().max() <= 1.0
    # not bit-equal: the 1x1x1 conv on all T latents at once is a plain GEMM whose kernel choice (256x256 rounds + 128x128
    # remainder rows) depends on T; the two tilings differ in the last fp32 bits, which the bf16 activations then amplify
    assert rel_l2(head, full[:, :13]) < 2e-3 and (head - full[:, :13]).abs().max() < 5e-2


def test_full_5b_engine_two_independent_kernel_sets_agree():
    """All 30 blocks at L = 9460 on the product kernels (256x256 GEMM + row split, 8-wave attention + splits) against the same
    engine forced onto the independent kernels (128x128-tile GEMM, register-staged 4-wave attention): two implementations of
    every hot product, same bf16 operand rounding points, different tilings / schedules / summation orders."""
    from yume_amd import framepack
    from yume_amd.wan23.modules.model import WanModel
    cfg = dict(synth.CFG_5B)
    with torch.device(DEV):
        model = WanModel(**cfg)
    synth.randomize_module_(model, seed=6)
    model = model.to(torch.bfloat16).eval().requires_grad_(False)
    F, H, W, lfz = 13, 44, 80, 8
    plan = framepack.pack_plan(F, H, W, lfz)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(48, F, H, W, device=DEV, generator=g)
    ctx = torch.randn(77, 4096, device=DEV, generator=g)
    t = torch.cat([torch.zeros(plan.n_hist_tok), torch.full((plan.n_new_tok,), 450.0)]).unsqueeze(0).to(DEV)

    def run():
        return model([x], t=t, context=[ctx], seq_len=plan.seq_len, latent_frame_zero=lfz, flag=True)[0]
    prod = run()
    model.engine.gemm_variant, model.engine.attn_variant = 1, 1
    alt = run()
    model.engine.gemm_variant, model.engine.attn_variant = 0, 0
    e = rel_l2(prod, alt)
    print(f"full 5B, 30 blocks: product kernels vs independent kernels rel-L2 {e:.3e}")
    assert torch.isfinite(prod).all() and e <= 1e-2
