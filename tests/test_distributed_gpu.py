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


def _one_rank_worker(rank, world, port, out_dir):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from yume_amd import distributed as D, synth
    from yume_amd.wan23.modules.model import WanModel
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)               # RCCL communicator of one rank on the test box's one GPU
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    cfg = synth.tiny_cfg("wan23")
    with torch.device(dev):
        m = WanModel(**cfg)
    synth.randomize_module_(m, seed=10)
    m = m.to(torch.bfloat16)
    before = [p.detach().clone() for p in m.parameters()]
    n1 = D.broadcast_module_(m, src=0, bucket_bytes=1 << 20, force=True)                 # 1 MiB device buckets (a larger tensor is its own) -> dist.broadcast each
    assert n1 >= 8, n1
    n2 = D.broadcast_module_(m, src=0, bucket_bytes=1 << 20, mode="scatter_allgather", force=True)   # dist.scatter + all_gather_into_tensor
    assert n2 == 2 * n1
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, m.parameters()))                # the round trip through the flat buckets is exact
    lat = torch.randn((48, 8, 12, 16), device=dev)
    allr = D.all_gather_results(lat, force=True)                                         # dist.all_gather on a device tensor
    assert allr.shape == (1, 48, 8, 12, 16) and torch.equal(allr[0], lat)
    chk = D.gather_scalars(3.25, device=dev, force=True)
    assert chk == [3.25]
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                             # bench.py's max-over-ranks timing collective
    dist.barrier()
    assert float(t) == 1.5
    with open(os.path.join(out_dir, "ok"), "w") as f:
        f.write(f"{n1} {n2}")
    dist.destroy_process_group()


def test_one_rank_rccl_communicator_drives_every_collective_of_the_multi_gpu_path(tmp_path):
    """no multi-GPU node exists for the builder, but the RCCL calls of the N > 1 path — communicator creation, the bucketed
    weight broadcast in both modes, the result all-gather, the scalar gather, the timing all-reduce and the barrier — run here through a
    real `backend="nccl"` group of ONE rank on device tensors, so that the driver's first N = 8 run is not also their first contact with
    RCCL (fastvideo/sample/sample_5b.py:1124-1134 is the launch being mirrored)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_one_rank_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok").exists()


def _run_bench(extra_args, env_extra, timeout=900):
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_8_launches_eight_ranks_itself(tmp_path):
    """`python bench.py --gpus 8` without torchrun's env must BECOME the 8-rank job the driver's SCALE run launches (the reference starts one
    process per GPU: fastvideo/sample/sample_5b.py:1124-1134, index = (step-1)*world + rank at :782-785). The test box has one device, so the
    eight ranks share cuda:0 and talk gloo (YUME_BENCH_SHARE_GPU=1, test mode); everything else is bench's own N = 8 control flow: rank 0
    builds the library while seven wait at the barrier, eight model builds, weight replication from rank 0, per-rank chains,
    barrier-bracketed timing, max over ranks, one JSON line from rank 0 — and the host-side rules of DESIGN §6: every rank caps its thread
    team at cpu_count // 8 (<= 16), and rank 0's CPU legs (cpu_baseline, parity) start only after every rank's GPU work is over."""
    import json
    r, line = _run_bench(["--gpus", "8", "--layers", "1", "--steps", "2", "--warmup", "1", "--no-vae"],
                         {"YUME_BENCH_SHARE_GPU": "1", "YUME_BENCH_RANK_LOG": str(tmp_path)}, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert line is not None, r.stdout[-2000:]
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert len(line["chain_checksums"]) == 8 and len(set(line["chain_checksums"])) == 8          # eight different chains
    assert line["weight_broadcast_collectives"] > 0
    assert line["config"]["parallelism"].startswith("dp8")
    assert line["value"] > 0 and abs(line["value"] - 8 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    cap = max(1, min(16, (os.cpu_count() or 8) // 8))
    assert line["host_threads_per_rank"] == [cap] * 8
    recs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(8)]
    assert sorted(x["rank"] for x in recs) == list(range(8)) and len({x["pid"] for x in recs}) == 8
    assert all(x["threads"] == cap for x in recs)
    assert line["cpu_legs_started_at"] >= max(x["gpu_work_done_at"] for x in recs)
    # an N > 1 line is complete: rank 0 emits the CPU baseline and the parity figure at any world size
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["host_threads"] >= line["cpu_baseline"]["cores"]
    assert line["parity"]["block"]["rel_l2"] <= 1e-2


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="a multi-GPU box runs it for real")
def test_bench_gpus_2_on_one_gpu_refuses_by_name():
    r, line = _run_bench(["--gpus", "2", "--layers", "1", "--steps", "1", "--warmup", "1", "--no-vae", "--no-cpu-baseline"], {}, timeout=300)
    assert r.returncode != 0 and line is None
    assert "--gpus 2 requested but this node shows 1 GPU" in r.stderr


def test_bench_gpus_mismatch_with_launcher_world_size_is_an_error():
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
