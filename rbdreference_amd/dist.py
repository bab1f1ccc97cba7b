"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

Every row of the batch is an independent evaluation (the reference keeps no state between calls,
``/root/reference/RBDReference.py:562-566, 656-662, 1132-1134``), so the path shards with NO
data-path collective: rank r evaluates rows ``shard_bounds(B, world, r)`` with the same per-robot
HIP library.  Collectives appear only at the edges and only when asked for:

* ``check_same_model``  -- all ranks must hold the same robot (hash all-gather, a few bytes);
* ``scatter_rows``      -- inputs that originate on one rank (grouped send/recv under RCCL);
* ``all_gather_rows``   -- outputs wanted on every rank; the default is to leave them sharded,
  which is what an MPC-style consumer of ``dc_du`` wants.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): an all-gather of cfg-3's 411 MB of ``dc_du``
costs ~2.3 ms as a ring but is not on the default path at all.
"""
from __future__ import annotations

import contextlib
from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "shard_sizes", "check_same_model", "scatter_rows", "all_gather_rows",
           "ShardedRBD"]


def shard_sizes(B: int, world: int) -> list:
    """Contiguous row blocks; the first ``B % world`` ranks get one extra row."""
    if B < 0 or world < 1:
        raise ValueError("B >= 0 and world >= 1 required")
    base, extra = divmod(B, world)
    return [base + (1 if r < extra else 0) for r in range(world)]


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int]:
    sizes = shard_sizes(B, world)
    start = sum(sizes[:rank])
    return start, start + sizes[rank]


def _world(group) -> Tuple[int, int]:
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def check_same_model(model_hash: str, group=None) -> None:
    world, _ = _world(group)
    if world == 1:
        return
    hashes = [None] * world
    dist.all_gather_object(hashes, model_hash, group=group)
    if len(set(hashes)) != 1:
        raise RuntimeError(f"ranks hold different robots: {hashes}")


def _global_rank(group, r: int) -> int:
    """Group-local rank -> the global rank ``dist.isend`` / ``dist.recv`` expect as peer."""
    return r if group is None else dist.get_global_rank(group, r)


def scatter_rows(x: Optional[torch.Tensor], B: int, row_shape: Sequence[int], dtype, device,
                 src: int = 0, group=None) -> torch.Tensor:
    """Rank ``src`` (a rank OF ``group``) holds ``x`` = [B, *row_shape]; every rank of the group
    receives its contiguous shard (point-to-point sends: xGMI is a full mesh, every shard travels
    over its own link)."""
    world, rank = _world(group)
    if world == 1:
        return x
    sizes = shard_sizes(B, world)
    out = torch.empty((sizes[rank], *row_shape), dtype=dtype, device=device)
    if rank == src:
        chunks = list(torch.split(x.contiguous(), sizes, dim=0))
        reqs = []
        for r in range(world):
            if r == src:
                out.copy_(chunks[r])
            elif sizes[r] > 0:
                reqs.append(dist.isend(chunks[r].contiguous(), dst=_global_rank(group, r), group=group))
        for q in reqs:
            q.wait()
    elif sizes[rank] > 0:
        dist.recv(out, src=_global_rank(group, src), group=group)
    return out


def all_gather_rows(local: torch.Tensor, B: int, group=None) -> torch.Tensor:
    """Concatenate the ranks' row shards (uneven shards are padded to the largest one)."""
    world, rank = _world(group)
    if world == 1:
        return local
    sizes = shard_sizes(B, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: shard has {local.shape[0]} rows, expected {sizes[rank]}")
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.zeros((mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    buf = torch.empty((world * mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.device.type == "cuda":           # RCCL: one collective into the flat buffer
        dist.all_gather_into_tensor(buf, pad.contiguous(), group=group)
    else:                                     # gloo (CPU tests): list form
        dist.all_gather(list(buf.split(mx, dim=0)), pad.contiguous(), group=group)
    if all(s == mx for s in sizes):
        return buf
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


class ShardedRBD:
    """Evaluate a GLOBAL batch that is replicated on (or scattered from) the ranks.

    ``compute`` maps local row shards to local outputs; by default it is the HIP-backed
    ``RBDReference`` of this rank.  (Tests inject a stand-in to exercise the sharding logic on CPU
    with the gloo backend -- the product path has no CPU implementation.)

    Shards are evaluated under ``rbd.shard_of(B)``: kernel selection sees the GLOBAL batch size, so the rows a
    rank returns are bit-identical to the same rows of an unsharded call whatever the number of ranks (without it
    a 65 536-row batch over 8 ranks would put 8 192-row shards on the small-batch column kernel).  For the same
    reason the constructor waits for the robot's own full library (``RbdLibrary.wait_specialized``): ranks must not
    mix the model-handle library's kernels (first-use path, equal to rounding only) with the specialised ones."""

    def __init__(self, rbd=None, group=None, compute_rnea_grad: Optional[Callable] = None,
                 compute_minv: Optional[Callable] = None, model_hash: Optional[str] = None):
        self.rbd = rbd
        self.group = group
        self._grad = compute_rnea_grad or (lambda q, qd, qdd, **kw: rbd.rnea_grad(q, qd, qdd, **kw))
        self._minv = compute_minv or (lambda q, **kw: rbd.minv(q, **kw))
        self._pin = rbd.shard_of if (rbd is not None and hasattr(rbd, "shard_of")) else (lambda B: contextlib.nullcontext())
        lib = getattr(rbd, "_lib", None)
        if lib is not None and hasattr(lib, "wait_specialized"):
            lib.wait_specialized()
        check_same_model(model_hash or (rbd.model.hash if rbd is not None else ""), group)

    def local_slice(self, B: int) -> slice:
        world, rank = _world(self.group)
        a, b = shard_bounds(B, world, rank)
        return slice(a, b)

    def rnea_grad(self, q, qd, qdd=None, gather: bool = False, **kw):
        """q, qd, qdd: the global [B, n] batch (same on every rank).  Returns this rank's rows of
        dc_du, or all rows when ``gather=True``."""
        B = q.shape[0]
        sl = self.local_slice(B)
        with self._pin(B):
            out = self._grad(q[sl], qd[sl], None if qdd is None else qdd[sl], **kw)
        if not gather:
            return out
        if isinstance(out, tuple):            # return_c=True -> (c, dc_du): gather each
            return tuple(all_gather_rows(o, B, self.group) for o in out)
        return all_gather_rows(out, B, self.group)

    def minv(self, q, gather: bool = False, **kw):
        B = q.shape[0]
        sl = self.local_slice(B)
        with self._pin(B):
            out = self._minv(q[sl], **kw)
        return all_gather_rows(out, B, self.group) if gather else out
