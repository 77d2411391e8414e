"""CPU, world_size 2, gloo: the N>1 sampling path - image sharding, the one text-feature broadcast, the latent
gather and the max-over-ranks timing reduction (lumina-t2x_amd/parallel.py)."""
import os
import socket

import torch
import torch.multiprocessing as mp

import lumina_t2x_amd  # noqa: F401
from lumina_t2x_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    n_img, T, C = 5, 16, 32
    feats = mask = None
    if rank == 0:
        g = torch.Generator().manual_seed(3)
        feats = torch.randn(n_img, 2, T, C, generator=g).to(torch.bfloat16)
        mask = (torch.rand(n_img, 2, T, generator=g) > 0.3).int()
    feats, mask = parallel.broadcast_prompts(feats, mask, src=0, device=torch.device("cpu"))
    g = torch.Generator().manual_seed(3)
    want = torch.randn(n_img, 2, T, C, generator=g).to(torch.bfloat16)
    assert torch.equal(feats, want) and mask.dtype == torch.int32
    mine = parallel.shard_range(n_img, rank, world)
    # stand-in for the per-image denoising: latent_i = mean of its cond features + image index
    local = torch.stack([feats[i, 0].float().mean().expand(4, 2, 2) + i for i in mine]) if len(mine) else torch.zeros(0, 4, 2, 2)
    full = parallel.gather_latents(local, n_img, dst=0)
    slow = parallel.max_over_ranks(1.0 + rank, torch.device("cpu"))
    assert slow == float(world)
    rep = parallel.comm_report(10.0 + rank, torch.device("cpu"))  # what bench.py records for N > 1
    assert rep["backend"] == "gloo" and rep["world"] == world and rep["ranks_seen"] == world
    assert rep["per_rank"] == [10.0 + r for r in range(world)] and rep["devices"] == ["cpu"] * world
    parallel.barrier()
    if rank == 0:
        expect = torch.stack([want[i, 0].float().mean().expand(4, 2, 2) + i for i in range(n_img)])
        assert torch.equal(full, expect)
        open(os.path.join(out_dir, "ok"), "w").write("1")
    else:
        assert full is None
    torch.distributed.destroy_process_group()


def test_two_rank_prompt_broadcast_and_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_range_partitions():
    for n in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in parallel.shard_range(n, r, world)]
            assert got == list(range(n))
            sizes = [len(parallel.shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
