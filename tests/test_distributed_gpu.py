"""2-rank RCCL smoke (backend "nccl" on ROCm): weight replication in both modes, result gather and two independent tiny denoise
chains, one process per GPU. Needs two GPUs — skipped on the single-GPU test box; the N > 1 logic itself is covered on CPU with
gloo (tests/test_distributed_cpu.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from yume_amd import distributed as D, framepack, synth
    from yume_amd.wan23.modules.model import WanModel
    r, w, local = D.init_from_env("nccl")
    dev = torch.device("cuda", local)
    cfg = synth.tiny_cfg("wan23")
    with torch.device(dev):
        m = WanModel(**cfg)
    synth.randomize_module_(m, seed=10 + rank)                   # ranks start with different weights
    n1 = D.broadcast_module_(m, src=0, bucket_bytes=1 << 20)
    with torch.device(dev):
        m2 = WanModel(**cfg)
    synth.randomize_module_(m2, seed=20 + rank)
    n2 = D.broadcast_module_(m2, src=0, bucket_bytes=1 << 20, mode="scatter_allgather")
    assert n2 == 2 * n1
    chk = D.gather_scalars(float(sum(p.double().abs().sum() for p in m.parameters())), device=dev)
    assert abs(chk[0] - chk[1]) == 0.0
    F, H, W, lfz = 13, 12, 16, 8
    plan = framepack.pack_plan(F, H, W, lfz)
    inp = synth.make_dit_inputs(cfg, "wan23", F, H, W, n_text=20, seed=100 + rank)      # its own prompt / noise
    t = torch.cat([torch.zeros(plan.n_hist_tok), torch.full((plan.n_new_tok,), 500.0)]).unsqueeze(0).double()
    out = m.eval()([inp["x"].to(dev)], t=t.to(dev), context=[inp["context"].to(dev)], seq_len=plan.seq_len, latent_frame_zero=lfz, flag=True)[0]
    allr = D.all_gather_results(out)
    assert allr.shape[0] == world and torch.isfinite(allr).all() and not torch.equal(allr[0], allr[1])
    torch.save(allr.cpu(), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL); the single-GPU box skips it")
def test_two_rank_rccl_replication_and_independent_chains(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a, b)                                     # both ranks gathered the same pair of results
