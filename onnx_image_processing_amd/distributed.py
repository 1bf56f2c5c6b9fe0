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


def env_world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str | None = None, force: bool = False) -> tuple[int, int, int]:
    """Join the process group described by the environment; no-op for a single process (unless `force`: a world of one
    still forms a group -- what the one-GPU hardware test of the RCCL code path uses)."""
    rank, world, local = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:          # a forced group of one outside any launcher: any free port
            import socket
            with socket.socket() as sck:
                sck.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sck.getsockname()[1])
        # the pool's host driver supports only dmabuf IPC: with the legacy mode RCCL's peer-buffer exchange fails in
        # hipIpcGetMemHandle ("invalid argument").  The image exports this already; set here for a bare environment.
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


def shard_sizes(total: int, world: int) -> list[int]:
    """Rows every rank holds under shard_range(total, rank, world)."""
    return [e - b for b, e in (shard_range(total, r, world) for r in range(world))]


def gather_records(rec: torch.Tensor, dst: int = 0, total: int | None = None,
                   collective: str = "gather") -> torch.Tensor | None:
    """Gather the per-rank record tensors to `dst`, concatenated in rank order (= global pair order under
    shard_range).  Returns None on the other ranks.

    total: the global number of pairs when the shards come from shard_range(total, rank, world) and may
    differ by one row: every rank pads its slab to the largest shard (the collective needs equal shapes; the
    padding is at most one record row per rank) and `dst` trims each slab back.  None: the caller guarantees
    equal shapes on every rank.
    collective: "gather" (each peer sends its slab straight to `dst`: one xGMI link each, no ring) or
    "all_gather" (every rank receives everything; for backends without gather).  It is a parameter, chosen
    identically on every rank by the caller -- never switched on a rank-local error, which would leave the
    ranks issuing different collectives."""
    if collective not in ("gather", "all_gather"):
        raise ValueError(f"collective must be 'gather' or 'all_gather', got {collective!r}")
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if total is not None and rec.shape[0] != total:
            raise ValueError(f"single process holds {rec.shape[0]} rows but total={total}")
        return rec
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = None
    if total is not None:
        sizes = shard_sizes(total, world)
        if rec.shape[0] != sizes[rank]:
            raise ValueError(f"rank {rank} holds {rec.shape[0]} rows, shard_range({total}, {rank}, {world}) says {sizes[rank]}")
        rows = max(sizes)
        if rec.shape[0] < rows:
            pad = torch.zeros((rows - rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
            rec = torch.cat([rec, pad], dim=0)
    rec = rec.contiguous()
    if collective == "gather":
        out = [torch.empty_like(rec) for _ in range(world)] if rank == dst else None
        dist.gather(rec, out, dst=dst)
        if rank != dst:
            return None
    else:
        full = torch.empty((world * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(full, rec)               # concatenated along dim 0 (the form gloo and RCCL share)
        if rank != dst:
            return None
        out = list(full.split(rec.shape[0], dim=0)) if rec.shape[0] > 0 else [rec] * world
    if sizes is not None:
        out = [slab[:sz] for slab, sz in zip(out, sizes)]
    return torch.cat(out, dim=0)


class PendingGather:
    """Handle of gather_records_async: wait() returns what gather_records would have returned (the concatenated records
    on `dst`, None elsewhere).  The tensors of the collective are kept alive until then."""

    def __init__(self, work, out, rank, dst, sizes, rec):
        self._work, self._out, self._rank, self._dst, self._sizes, self._rec = work, out, rank, dst, sizes, rec

    def wait(self) -> torch.Tensor | None:
        if self._work is not None:
            self._work.wait()          # RCCL: the current stream waits for the collective; the host does not block
            self._work = None
        if self._rank != self._dst:
            return None
        out = self._out
        if self._sizes is not None:
            out = [slab[:sz] for slab, sz in zip(out, self._sizes)]
        return out[0] if len(out) == 1 else torch.cat(out, dim=0)


def gather_records_async(rec: torch.Tensor, dst: int = 0, total: int | None = None) -> PendingGather:
    """gather_records(collective="gather") issued WITHOUT waiting for it: the collective runs on the backend's own
    stream behind the work already enqueued on the current stream (the records), so the next step's kernels can be
    enqueued -- and run -- while the slabs travel over xGMI.  bench.py keeps one gather in flight: step i waits for the
    gather of step i - 1.  Same argument checks and padding rules as gather_records."""
    if not dist.is_initialized():
        if total is not None and rec.shape[0] != total:
            raise ValueError(f"single process holds {rec.shape[0]} rows but total={total}")
        return PendingGather(None, [rec], 0, 0, None, rec)
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = None
    if total is not None:
        sizes = shard_sizes(total, world)
        if rec.shape[0] != sizes[rank]:
            raise ValueError(f"rank {rank} holds {rec.shape[0]} rows, shard_range({total}, {rank}, {world}) says {sizes[rank]}")
        rows = max(sizes)
        if rec.shape[0] < rows:
            pad = torch.zeros((rows - rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
            rec = torch.cat([rec, pad], dim=0)
    rec = rec.contiguous()
    out = [torch.empty_like(rec) for _ in range(world)] if rank == dst else None
    work = dist.gather(rec, out, dst=dst, async_op=True)
    return PendingGather(work, out, rank, dst, sizes, rec)


def barrier_max_ms(elapsed_ms: float, device: torch.device | str) -> float:
    """Maximum of a per-rank time over all ranks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed_ms
    t = torch.tensor([elapsed_ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
