"""One-process-per-GPU batch sharding for sampling (new functionality: every reference sampler asserts
``num_gpus == 1`` - lumina_next_t2i/sample.py:339).

Each image's ODE trajectory is independent, so images are partitioned over ranks and the only data-path
exchange is ONE broadcast of the text features (``cap_feats [n_img*2, T, C]`` + mask, <= a few MB) from
the rank that ran the text encoder, plus an optional gather of the final latents.  On ROCm the "nccl"
backend is RCCL; a root->peers broadcast of this size is latency bound on the direct xGMI links, so no
bucketing / ring tuning applies.  The same code runs on CPU with the gloo backend (tests).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, share_device: bool = False) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the default group if world > 1.

    ``share_device`` (or LUMINA_SHARE_DEVICE=1): every rank drives GPU 0 and the collectives run over gloo with a host hop (RCCL
    cannot put two ranks on one device).  It exists to run the REAL engine under the real launcher / rank / shard / relay logic on a
    one-GPU lease; it says nothing about RCCL, xGMI or scaling - the returned local rank is 0 for every rank."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    share_device = share_device or os.environ.get("LUMINA_SHARE_DEVICE", "0") == "1"
    if share_device:
        local, backend = 0, "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local if local < torch.cuda.device_count() else 0)
            dist.init_process_group(backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local if local < torch.cuda.device_count() else 0)
    return rank, world, local


def _host_hop() -> bool:
    """gloo moves host memory: device tensors take a D2H / H2D hop around the collective (the CPU twin of the RCCL path, and the
    shared-device mode of init_distributed)"""
    return dist.get_backend() == "gloo"


def _broadcast(t: torch.Tensor, src: int) -> None:
    if t.is_cuda and _host_hop():
        h = t.cpu()
        dist.broadcast(h, src)
        t.copy_(h)
    else:
        dist.broadcast(t, src)


def shard_range(n_items: int, rank: int, world: int) -> range:
    """contiguous, balanced partition: the first n_items % world ranks get one extra item"""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def broadcast_prompts(cap_feats: Optional[torch.Tensor], cap_mask: Optional[torch.Tensor], *, src: int = 0,
                      device: Optional[torch.device] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rank ``src`` holds cap_feats [n_img, 2, T, C] (cond, uncond per image) and cap_mask [n_img, 2, T]; every
    rank returns the full tensors.  Shapes/dtype travel first as a tiny int64 header."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return cap_feats, cap_mask
    rank = dist.get_rank()
    dev = device or (cap_feats.device if cap_feats is not None else
                     (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")))
    dtypes = [torch.float32, torch.bfloat16, torch.float16]
    if rank == src:
        hdr = torch.tensor(list(cap_feats.shape) + [dtypes.index(cap_feats.dtype)], dtype=torch.int64, device=dev)
    else:
        hdr = torch.zeros(5, dtype=torch.int64, device=dev)
    _broadcast(hdr, src)
    n, two, T, C, di = (int(v) for v in hdr.tolist())
    if rank != src:
        cap_feats = torch.empty(n, two, T, C, dtype=dtypes[di], device=dev)
        cap_mask = torch.empty(n, two, T, dtype=torch.int32, device=dev)
    else:
        cap_feats = cap_feats.to(dev).contiguous()
        cap_mask = cap_mask.to(device=dev, dtype=torch.int32).contiguous()
    _broadcast(cap_feats, src)
    _broadcast(cap_mask, src)
    return cap_feats, cap_mask


def gather_latents(local: torch.Tensor, n_items: int, *, dst: int = 0) -> Optional[torch.Tensor]:
    """inverse of shard_range: rank ``dst`` gets [n_items, ...] in item order, others None"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [len(shard_range(n_items, r, world)) for r in range(world)]
    cap = max(counts)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out_dev = pad.device
    if pad.is_cuda and _host_hop():
        pad = pad.cpu()
    bufs: Optional[List[torch.Tensor]] = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0).to(out_dev)


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if _host_hop() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def comm_report(per_rank_value: float, device: torch.device) -> dict:
    """What a multi-rank bench line records so that the JSON itself proves the collective layer saw every rank:
    the backend in use, the RCCL version torch was built against, `ranks_seen` = an all-reduce SUM of ones (== world only if every
    rank took part in a real collective), every rank's own value of `per_rank_value` (all-gather, rank order) and device name."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return {"backend": None, "world": 1, "ranks_seen": 1, "per_rank": [float(per_rank_value)],
                "devices": [torch.cuda.get_device_name(device) if device.type == "cuda" else "cpu"]}
    world, host = dist.get_world_size(), _host_hop()
    cdev = torch.device("cpu") if host else device
    ones = torch.ones(1, dtype=torch.int64, device=cdev)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    mine = torch.tensor([per_rank_value], dtype=torch.float64, device=cdev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    name = torch.cuda.get_device_name(device) if device.type == "cuda" else "cpu"
    idx = device.index if device.type == "cuda" and device.index is not None else -1
    names = [None] * world
    dist.all_gather_object(names, f"{name} (cuda:{idx})" if device.type == "cuda" else name)
    rccl = None
    try:
        if dist.get_backend() == "nccl":
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl = None
    return {"backend": dist.get_backend(), "rccl_version": rccl, "world": world, "ranks_seen": int(ones.item()),
            "per_rank": [float(v.item()) for v in every], "devices": names}


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
