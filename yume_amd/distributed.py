"""Multi-GPU sampling: one process per GPU, independent denoise chains, replicated weights.

The reference's only multi-GPU mode at inference is data parallel over samples — `index = (step-1)*world_size + rank`
(fastvideo/sample/sample_5b.py:782-785) — with the DiT wrapped in FSDP FULL_SHARD, i.e. a parameter all-gather per block
per forward that exists only because of 80 GB GPUs. On MI355X (288 GB HBM3E) the weights are replicated, so there is NO
collective inside the step loop; RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests) is used for
  * broadcast_module_: one-time weight replication from rank 0 in few, large flat buckets (xGMI rings are per-link bound,
    so bucket size rather than message count is what matters); mode "scatter_allgather"  sends each peer
    a DIFFERENT 1/world slice of the bucket — rank 0's seven xGMI links each carry 1/8 of the bytes instead of one ring
    carrying all of them — and completes the replicas with one all-gather in which every link of every GPU works;
  * all_gather_results / gather_scalars: result latents, checksums and timings at chunk end.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def shard_indices(n_items, rank, world):
    """items processed by `rank`: index = (step-1)*world + rank for step = 1, 2, ... (sample_5b.py:782-785)."""
    return list(range(rank, n_items, world))


@torch.no_grad()
def broadcast_module_(module, src=0, bucket_bytes=1 << 30, mode=None, force=False):
    """Replicate rank `src`'s parameters and buffers on every rank in flat buckets. mode "broadcast" = one dist.broadcast per
    bucket; "scatter_allgather" = dist.scatter of world slices + dist.all_gather_into_tensor (2 collectives per bucket; the
    bucket is padded to a multiple of world). Default: YUME_WEIGHT_DIST env var, else "broadcast". Returns the collective count.
    force: issue the collectives in a one-rank group too (tests/test_distributed_gpu.py drives the RCCL calls on a single GPU)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return 0
    mode = mode or os.environ.get("YUME_WEIGHT_DIST", "broadcast")
    if mode not in ("broadcast", "scatter_allgather"):
        raise ValueError(f"unknown weight distribution mode {mode!r}")
    world, rank = dist.get_world_size(), dist.get_rank()
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    n_coll = 0
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    host_staged = dist.get_backend() == "gloo"      # gloo moves host memory: device buckets are staged through the host
    for (dtype, pdev), ts in by_dtype.items():
        dev = torch.device("cpu") if host_staged else pdev
        cap = max(1, bucket_bytes // ts[0].element_size())
        i = 0
        while i < len(ts):
            group, n = [], 0
            while i < len(ts) and (not group or n + ts[i].numel() <= cap):
                group.append(ts[i])
                n += ts[i].numel()
                i += 1
            npad = (n + world - 1) // world * world if mode == "scatter_allgather" else n
            flat = torch.empty(npad, dtype=dtype, device=dev)
            off = 0
            if rank == src:
                for t in group:
                    flat[off:off + t.numel()].copy_(t.reshape(-1))
                    off += t.numel()
                flat[n:].zero_()
            if mode == "broadcast":
                dist.broadcast(flat, src=src)
                n_coll += 1
            else:
                piece = torch.empty(npad // world, dtype=dtype, device=dev)
                dist.scatter(piece, list(flat.view(world, -1).unbind(0)) if rank == src else None, src=src)
                dist.all_gather_into_tensor(flat, piece)
                n_coll += 2
            off = 0
            for t in group:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
    return n_coll


def all_gather_results(t, force=False):
    """[world, *t.shape]: every rank's result tensor (e.g. the [48,8,44,80] latents of its chain)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return t.unsqueeze(0)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t.contiguous())
    return torch.stack(out)


def gather_scalars(x, device=None, force=False):
    """list of one python float per rank (timings, checksums)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return [float(x)]
    dev = device or (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    return [float(v) for v in all_gather_results(t, force).flatten()]
