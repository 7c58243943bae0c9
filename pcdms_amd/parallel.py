"""Data-parallel harness for the stage-2 sampler: one process per GPU, pairs sharded, ONE all-gather.

The reference shards the test pairs over processes with no communication at all
(``split_list_into_chunks`` + one ``mp.Process`` per ``cuda:{rank}``,
/root/reference/stage2_batchtest_inpaint_model.py:25-31,266-285) and each process writes its own PNGs.
Here every rank samples its chunk with replicated weights and a single collective
(``all_gather`` -- RCCL over xGMI with backend "nccl", gloo in the CPU tests) returns every pair's
final latents to every rank in the original pair order.  There is no collective inside the denoise
loop; the payload is <= 0.36 MB of fp32 latents per sample, i.e. latency-bound.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, TypeVar

import torch
import torch.distributed as dist

T = TypeVar("T")


def split_list_into_chunks(lst: Sequence[T], n: int) -> List[List[T]]:
    """Static block partition with the remainder folded into the last chunk -- the reference's
    partition (stage2_batchtest_inpaint_model.py:25-31) -- but total for every n >= 1 (the reference
    raises for len(lst) < n): always returns exactly n chunks, some possibly empty."""
    if n <= 0:
        raise ValueError("n must be positive")
    size = len(lst) // n
    if size == 0:
        return [[x] for x in lst] + [[] for _ in range(n - len(lst))]
    chunks = [list(lst[i * size:(i + 1) * size]) for i in range(n)]
    chunks[-1].extend(lst[n * size:])
    return chunks


def chunk_index_ranges(total: int, n: int) -> List[range]:
    out, start = [], 0
    for c in split_list_into_chunks(list(range(total)), n):
        out.append(range(start, start + len(c)))
        start += len(c)
    return out


def run_sharded(pairs: Sequence[T], sample_fn: Callable[[T], torch.Tensor], group: Optional[dist.ProcessGroup] = None
                ) -> List[torch.Tensor]:
    """``sample_fn(pair) -> Tensor`` (same shape/dtype for every pair) is applied to this rank's chunk; returns
    the results of ALL pairs, in pair order, on every rank.  Exactly one collective."""
    if not dist.is_initialized():
        return [sample_fn(p) for p in pairs]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    chunks = split_list_into_chunks(list(pairs), world)
    mine = [sample_fn(p) for p in chunks[rank]]
    longest = max(len(c) for c in chunks)
    if longest == 0:
        return []
    proto = mine[0] if mine else None
    if proto is None:  # empty chunk: learn shape/dtype from the sampler's declared example
        ex = getattr(sample_fn, "example_output", None)
        if ex is None:
            raise ValueError("rank with an empty chunk needs sample_fn.example_output (a tensor of the result shape)")
        proto = ex
    send = torch.zeros((longest,) + tuple(proto.shape), dtype=proto.dtype, device=proto.device)
    for i, t in enumerate(mine):
        send[i].copy_(t)
    recv = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(recv, send, group=group)
    else:
        parts = list(recv.unbind(0))
        dist.all_gather(parts, send, group=group)
    out: List[torch.Tensor] = []
    for r, c in enumerate(chunks):
        out.extend(recv[r, i] for i in range(len(c)))
    return out
