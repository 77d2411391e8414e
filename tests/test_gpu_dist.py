"""-m gpu: the N > 1 sampling path - over RCCL (backend "nccl") - runs when the box exposes >= 2 GPUs, skips otherwise (the
authoring leases are single-GPU; the CPU twin of this test is tests/test_dist_gloo.py, world size 2 over gloo).

1. parallel.broadcast_prompts / gather_latents / max_over_ranks across 2 ranks, one per GPU;
2. `python bench.py --gpus 2` exactly as the driver invokes it for N = 1 (no torchrun, no WORLD_SIZE in the environment): bench.py
   must launch its own ranks and rank 0 must print one JSON line with n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

import lumina_t2x_amd  # noqa: F401
from lumina_t2x_amd import parallel

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL over xGMI)")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r, w, local = parallel.init_distributed("nccl")
    dev = torch.device("cuda", local)
    n_img, T, C = 5, 16, 64
    feats = mask = None
    if rank == 0:
        g = torch.Generator().manual_seed(3)
        feats = torch.randn(n_img, 2, T, C, generator=g).to(dev, torch.bfloat16)
        mask = (torch.rand(n_img, 2, T, generator=g) > 0.3).int().to(dev)
    feats, mask = parallel.broadcast_prompts(feats, mask, src=0, device=dev)
    g = torch.Generator().manual_seed(3)
    want = torch.randn(n_img, 2, T, C, generator=g).to(torch.bfloat16)
    assert feats.device == dev and torch.equal(feats.cpu(), want) and mask.dtype == torch.int32
    mine = parallel.shard_range(n_img, rank, world)
    local_lat = torch.stack([feats[i, 0].float().mean().expand(4, 2, 2) + i for i in mine]) if len(mine) else torch.zeros(0, 4, 2, 2, device=dev)
    full = parallel.gather_latents(local_lat, n_img, dst=0)
    assert parallel.max_over_ranks(1.0 + rank, dev) == float(world)
    parallel.barrier()
    if rank == 0:
        expect = torch.stack([want[i, 0].float().mean().expand(4, 2, 2) + i for i in range(n_img)])
        assert torch.allclose(full.cpu(), expect)
        open(os.path.join(out_dir, "ok"), "w").write("1")
    torch.distributed.destroy_process_group()


@needs2
def test_two_rank_prompt_broadcast_and_gather_over_rccl(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


@needs2
def test_bench_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    # pre-flight of the driver's scaling run: WITHOUT --share-device the collectives are RCCL's, every rank took part in
    # one, and the two ranks sit on two different devices
    comm = rec["comm"]
    assert comm["backend"] == "nccl" and comm["rccl_version"], comm
    assert comm["world"] == 2 and comm["ranks_seen"] == 2 and len(comm["ms_per_step_per_rank"]) == 2, comm
    assert len(set(comm["devices"])) == 2 and all("cuda:" in d for d in comm["devices"]), comm["devices"]
    assert comm["collective_in_timed_region"] is False and "shared_device" not in rec


# ---- the same N > 1 code on ONE GPU: both ranks drive device 0, collectives over gloo with a host hop (parallel.init_distributed
#      share_device).  Runs on every lease; proves the launcher / rank / shard / relay logic with the real engine - not RCCL, not scaling.
def _worker_shared(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LUMINA_SHARE_DEVICE="1")
    r, w, local = parallel.init_distributed("nccl")  # the environment switch overrides the backend
    assert local == 0 and torch.distributed.get_backend() == "gloo" and torch.cuda.current_device() == 0
    dev = torch.device("cuda", 0)
    n_img, T, C = 3, 8, 32
    feats = mask = None
    if rank == 0:
        g = torch.Generator().manual_seed(9)
        feats = torch.randn(n_img, 2, T, C, generator=g).to(dev, torch.bfloat16)
        mask = (torch.rand(n_img, 2, T, generator=g) > 0.3).int().to(dev)
    feats, mask = parallel.broadcast_prompts(feats, mask, src=0, device=dev)
    g = torch.Generator().manual_seed(9)
    want = torch.randn(n_img, 2, T, C, generator=g).to(torch.bfloat16)
    assert feats.device == dev and torch.equal(feats.cpu(), want) and mask.dtype == torch.int32 and mask.device == dev
    mine = parallel.shard_range(n_img, rank, world)
    local_lat = torch.stack([feats[i, 0].float().mean().expand(4, 2, 2) + i for i in mine])
    full = parallel.gather_latents(local_lat, n_img, dst=0)
    assert parallel.max_over_ranks(1.0 + rank, dev) == float(world)
    parallel.barrier()
    if rank == 0:
        expect = torch.stack([want[i, 0].float().mean().expand(4, 2, 2) + i for i in range(n_img)])
        assert full.device == dev and torch.allclose(full.cpu(), expect)
        open(os.path.join(out_dir, "ok"), "w").write("1")
    torch.distributed.destroy_process_group()


def test_two_ranks_share_one_gpu_prompt_broadcast_and_gather(tmp_path):
    mp.spawn(_worker_shared, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_bench_launches_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 --share-device` as the driver would start an N > 1 run (no torchrun, no WORLD_SIZE): the script launches
    its own two ranks under torch.distributed.run, each builds the real engine on GPU 0 and denoises its own image, the prompt is
    broadcast from rank 0, the time is the max over ranks and rank 0 alone prints the JSON line - labelled as what it is."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LUMINA_SHARE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-device", "--steps", "3", "--warmup", "1",
                          "--repeats", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["ranks"] == 2 and rec["shared_device"] is True and rec["steps"] == 3 and rec["value"] > 0
    assert "NOT a scaling" in rec["note"]
    # round 5: the line itself shows what the collective layer saw - every rank took part in an all-reduce, each reports its own time
    comm = rec["comm"]
    assert comm["backend"] == "gloo" and comm["world"] == 2 and comm["ranks_seen"] == 2 and len(comm["ms_per_step_per_rank"]) == 2
    assert all(v > 0 for v in comm["ms_per_step_per_rank"]) and max(comm["ms_per_step_per_rank"]) == pytest.approx(rec["ms_per_step"], rel=1e-6)
    assert len(comm["devices"]) == 2 and comm["collective_in_timed_region"] is False


def _worker_rccl_one_rank(rank, world, port, out_dir):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # exactly the call parallel.init_distributed makes for world > 1 (backend "nccl" = RCCL, bound to the rank's device)
    torch.distributed.init_process_group("nccl", device_id=dev)
    t = torch.arange(8, device=dev, dtype=torch.float32)
    torch.distributed.broadcast(t, 0)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    h = torch.ones(3, 4, device=dev, dtype=torch.bfloat16)
    bufs = [torch.empty_like(h)]
    torch.distributed.gather(h, bufs, dst=0)
    torch.distributed.barrier()
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32)) and torch.equal(bufs[0], h)
    assert parallel.max_over_ranks(2.5, dev) == 2.5  # world 1: early return, no collective
    open(os.path.join(out_dir, "ok"), "w").write("1")
    torch.distributed.destroy_process_group()


def test_rccl_communicator_and_collectives_with_one_rank(tmp_path):
    """What a one-GPU lease CAN say about the RCCL path: the `nccl` backend initialises with `device_id` on this stack and the four
    collectives parallel.py uses (broadcast, all_reduce MAX, gather, barrier) run on device tensors - with ONE rank, i.e. no xGMI
    traffic and no peer.  It catches a missing RCCL library, an init signature the installed torch rejects, or an IPC-mode
    environment that breaks communicator creation; it is not a multi-GPU test."""
    mp.spawn(_worker_rccl_one_rank, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok").exists()
