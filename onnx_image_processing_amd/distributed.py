"""Multi-GPU layer: independent image pairs are sharded over ranks, one process per GPU.

The path has no exchange step during compute: a pair is a pure function of
its two images.  The only collective is the gather of fixed-size match records to rank 0 at
the end of a step -- `torch.distributed.gather`, which is RCCL over xGMI with backend "nccl"
on ROCm (each peer sends its slab straight to rank 0; ~2.4 KB per pair, far below one link).
The same code runs on CPU tensors with backend "gloo" (tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

RECORD_FIELDS = 6  # y1, x1, y2, x2, score, valid
_use_all_gather = False


def env_world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str | None = None) -> tuple[int, int, int]:
    """Join the process group described by the environment; no-op for a single process."""
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition of `total` units: [begin, end) for `rank`; sizes differ by <= 1."""
    if total < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def pack_records(mk1: torch.Tensor, mk2: torch.Tensor, scores: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    """(B,Mx,2),(B,Mx,2),(B,Mx),(B,Mx) -> one float32 record tensor (B, Mx, 6)."""
    return torch.cat([mk1, mk2, scores.unsqueeze(-1), valid.to(scores.dtype).unsqueeze(-1)], dim=-1).contiguous()


def unpack_records(rec: torch.Tensor):
    return rec[..., 0:2], rec[..., 2:4], rec[..., 4], rec[..., 5] > 0.5


def gather_records(rec: torch.Tensor, dst: int = 0) -> torch.Tensor | None:
    """Gather equally-shaped per-rank record tensors to `dst`, concatenated in rank order
    (= global pair order under shard_range with equal shard sizes).  Returns None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    world, rank = dist.get_world_size(), dist.get_rank()
    global _use_all_gather
    if not _use_all_gather:
        try:
            out = [torch.empty_like(rec) for _ in range(world)] if rank == dst else None
            dist.gather(rec, out, dst=dst)
            return torch.cat(out, dim=0) if rank == dst else None
        except RuntimeError:                      # a backend without gather: every rank takes all slabs instead
            _use_all_gather = True
    full = torch.empty((world,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(full, rec)
    return full.reshape((-1,) + tuple(rec.shape[1:])) if rank == dst else None


def barrier_max_ms(elapsed_ms: float, device: torch.device | str) -> float:
    """Maximum of a per-rank time over all ranks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed_ms
    t = torch.tensor([elapsed_ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
