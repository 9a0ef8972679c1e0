"""Ulysses exchange on CPU: world_size-2 and -4 `gloo` groups. Each rank starts with its token chunk of
q | k (token-major) and V^T (K-major) for ALL heads; after exchange_qkv it must hold ALL tokens of ITS heads in the layouts
the attention kernel takes; a per-rank exact-softmax attention followed by exchange_out must equal the single-process
attention over all heads. The device kernels are not involved: this covers the partition, packing and collectives."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT
from test_distributed_cpu import _free_port

sys.path.insert(0, ROOT)

D = 16          # head dim (the layout code is head-dim agnostic)


def _global(L, H, world):
    g = torch.Generator().manual_seed(11)
    C = H * D
    Lp = ((-(-L // world)) + 7) // 8 * 8
    Lt = world * Lp
    q = torch.randn(Lt, C, generator=g).to(torch.bfloat16)
    k = torch.randn(Lt, C, generator=g).to(torch.bfloat16)
    v = torch.randn(Lt, C, generator=g).to(torch.bfloat16)
    return q, k, v, Lp


def _attention(q, k, v, H, n_keys):
    Lq = q.shape[0]
    qh, kh, vh = (t.float().view(t.shape[0], H, D).transpose(0, 1) for t in (q, k, v))
    s = qh @ kh.transpose(1, 2) * D ** -0.5
    s[:, :, n_keys:] = float("-inf")
    return (torch.softmax(s, -1) @ vh).transpose(0, 1).reshape(Lq, H * D)


def _worker(rank, world, port, L, H):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yume_amd.ulysses import SequenceParallel
    sp = SequenceParallel()
    assert sp.host_staged and sp.rank == rank and sp.world == world
    sp.check_heads(H)
    q, k, v, Lp = _global(L, H, world)
    C = H * D
    Lp2, lo, hi = sp.chunk(L)
    assert Lp2 == Lp and lo == min(rank * Lp, L) and hi == min(lo + Lp, L)
    rows = slice(rank * Lp, (rank + 1) * Lp)
    qk_loc = torch.cat([q[rows], k[rows]], dim=1).contiguous()
    vt_loc = v[rows].t().contiguous()                                   # [C, Lp] K-major
    qf, kf, vtf = sp.exchange_qkv(qk_loc, vt_loc, C)
    cols = slice(rank * C // world, (rank + 1) * C // world)
    assert torch.equal(qf, q[:, cols]) and torch.equal(kf, k[:, cols])
    assert torch.equal(vtf, v[:, cols].t())
    of = _attention(qf, kf, vtf.t(), H // world, L).to(torch.bfloat16)
    o_loc = sp.exchange_out(of.contiguous())
    want = _attention(q, k, v, H, L).to(torch.bfloat16)
    assert torch.equal(o_loc, want[rows])
    y = sp.gather_rows(o_loc[:, :5].float().contiguous())
    assert torch.equal(y, want[:, :5].float())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,L,H", [(2, 37, 4), (4, 50, 8), (2, 16, 2)])
def test_ulysses_exchange_matches_single_process(world, L, H):
    mp.spawn(_worker, args=(world, _free_port(), L, H), nprocs=world, join=True)


def test_world_one_is_identity():
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from yume_amd.ulysses import SequenceParallel
        sp = SequenceParallel()
        q, k, v, Lp = _global(21, 2, 1)
        qf, kf, vtf = sp.exchange_qkv(torch.cat([q, k], 1).contiguous(), v.t().contiguous(), 2 * D)
        assert torch.equal(qf, q) and torch.equal(kf, k) and torch.equal(vtf, v.t())
        assert torch.equal(sp.exchange_out(q.contiguous()), q)
        with pytest.raises(RuntimeError):
            SequenceParallel.check_heads(type("S", (), {"world": 4})(), 6)
    finally:
        dist.destroy_process_group()


# ---- against the reference's own distributed_attention (wan23/distributed/ulysses.py:9-47), executed under gloo ----------------------
def _ref_worker(rank, world, port, L, H):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ref_import
    from yume_amd.ulysses import SequenceParallel
    ref = ref_import.ref_ulysses()
    sp = SequenceParallel()
    q, k, v, Lp = _global(L, H, world)
    assert world * Lp == L, "the reference chunks the sequence evenly: choose L = world * multiple of 8"
    C = H * D
    rows = slice(rank * Lp, (rank + 1) * Lp)
    # the reference: token-major [B, L/P, N, D] shards in, [B, L/P, N, D] out (4 list all-to-alls + flash_attention in the middle)
    shard = lambda t: t[rows].float().view(1, Lp, H, D)
    want = ref.distributed_attention(shard(q), shard(k), shard(v), seq_lens=torch.tensor([L])).reshape(Lp, C)
    # here: one packed exchange in, the same attention on the rank's heads, one exchange out
    qf, kf, vtf = sp.exchange_qkv(torch.cat([q[rows], k[rows]], dim=1).contiguous(), v[rows].t().contiguous(), C)
    got = sp.exchange_out(_attention(qf, kf, vtf.t(), H // world, L).contiguous())
    assert got.shape == want.shape and torch.allclose(got, want, rtol=0, atol=2e-6), (got - want).abs().max()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,L,H", [(2, 48, 4), (4, 64, 8)])
def test_exchange_matches_the_references_distributed_attention(world, L, H):
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("needs the reference tree (build container only)")
    mp.spawn(_ref_worker, args=(world, _free_port(), L, H), nprocs=world, join=True)
