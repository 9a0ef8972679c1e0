"""N > 1 path on CPU: world_size-2 `gloo` process group — rank sharding (index = (step-1)*world + rank), flat-bucketed
weight broadcast, result all-gather, and a sharded multi-prompt sampling run whose gathered results equal the
single-process results (no collective inside the step loop, so the chains must be bit-identical)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TinyNet(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.a = torch.nn.Parameter(torch.randn(6, 6, generator=g) * 0.3)
        self.b = torch.nn.Parameter(torch.randn(100003, generator=g))            # odd size: exercises bucket packing
        self.register_buffer("c", torch.randn(17, generator=g))
        self.h = torch.nn.Parameter(torch.randn(33, generator=g).to(torch.bfloat16))


def chain(net, prompt_seed, steps=5, lfz=2):
    from yume_amd import sampling
    g = torch.Generator().manual_seed(prompt_seed)
    hist = torch.randn(6, 3, 4, 5, generator=g)
    x = torch.randn(6, lfz, 4, 5, generator=g)
    sig = sampling.sampling_sigmas(steps, 7.0)
    vel = lambda lat, i: torch.tanh(torch.einsum("cd,dfhw->cfhw", net.a, lat)) + net.b[:1] * 0.01 * i
    return sampling.ode_chunk(vel, torch.cat([hist, x], 1), sig, lfz, sampling.clean_history(hist))[:, -lfz:]


def _worker(rank, world, port, n_prompts, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from yume_amd import distributed as D
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    net = TinyNet(seed=100 + rank)                     # ranks start with DIFFERENT weights
    n_coll = D.broadcast_module_(net, src=0, bucket_bytes=64 * 1024)
    ref = TinyNet(seed=100)
    for (k, p), (_, q) in zip(net.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(p, q), k
    assert n_coll >= 3                                  # fp32 split in >= 2 buckets + the bf16 bucket
    # the scatter + all-gather distribution (every link of rank 0 carries a different slice) replicates the same bits
    net2 = TinyNet(seed=200 + rank)
    n2 = D.broadcast_module_(net2, src=1, bucket_bytes=64 * 1024, mode="scatter_allgather")
    ref2 = TinyNet(seed=201)
    for (k, p), (_, q) in zip(net2.state_dict().items(), ref2.state_dict().items()):
        assert torch.equal(p, q), k
    assert n2 == 2 * n_coll
    mine = D.shard_indices(n_prompts, rank, world)
    assert mine == [i for i in range(n_prompts) if i % world == rank]
    with torch.no_grad():
        res = torch.stack([chain(net, 1000 + i) for i in mine])
    allres = D.all_gather_results(res)                  # [world, n/world, ...]
    times = D.gather_scalars(0.5 + rank)
    assert times == [0.5 + k for k in range(world)]
    if rank == 0:
        torch.save(allres, os.path.join(out_dir, "gathered.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_sampling(tmp_path):
    world, n_prompts = 2, 4
    mp.spawn(_worker, args=(world, _free_port(), n_prompts, str(tmp_path)), nprocs=world, join=True)
    gathered = torch.load(os.path.join(tmp_path, "gathered.pt"))
    net = TinyNet(seed=100)
    with torch.no_grad():
        for i in range(n_prompts):
            want = chain(net, 1000 + i)
            got = gathered[i % world, i // world]
            assert torch.equal(got, want), i


def test_single_process_helpers_are_noops():
    from yume_amd import distributed as D
    net = TinyNet(1)
    assert D.broadcast_module_(net) == 0
    t = torch.arange(6.).view(2, 3)
    assert torch.equal(D.all_gather_results(t), t.unsqueeze(0))
    assert D.gather_scalars(3.0) == [3.0]
    assert D.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]


def test_bench_gpus_n_without_enough_gpus_refuses_by_name():
    """bench.py --gpus 2 outside torchrun launches its own ranks (tests/test_distributed_gpu.py); on a node with fewer than 2 GPUs
    (this container: none) it must refuse by name instead of silently running one rank and printing n_gpus: 1."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "YUME_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    if torch.cuda.device_count() >= 2:
        pytest.skip("multi-GPU node")
    assert r.returncode != 0
    assert "--gpus 2 requested but this node shows" in r.stderr and "{" not in r.stdout


def _one_rank_worker(rank, world, port):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from yume_amd import distributed as D
    dist.init_process_group("gloo")
    net = TinyNet(seed=7)
    ref = TinyNet(seed=7)
    assert D.broadcast_module_(net) == 0                                    # a one-rank group has nothing to replicate ...
    n1 = D.broadcast_module_(net, bucket_bytes=64 * 1024, force=True)       # ... unless forced (the single-GPU RCCL test drives it so)
    n2 = D.broadcast_module_(net, bucket_bytes=64 * 1024, mode="scatter_allgather", force=True)
    assert n1 >= 3 and n2 == 2 * n1
    for (k, p), (_, q) in zip(net.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(p, q), k
    t = torch.randn(3, 4)
    assert torch.equal(D.all_gather_results(t, force=True), t.unsqueeze(0))
    assert D.gather_scalars(2.5, force=True) == [2.5]
    dist.destroy_process_group()


def test_forced_collectives_in_a_one_rank_group():
    """the `force` switch tests/test_distributed_gpu.py uses to run every RCCL call of the N > 1 path on the single-GPU box, here over gloo."""
    mp.spawn(_one_rank_worker, args=(1, _free_port()), nprocs=1, join=True)
