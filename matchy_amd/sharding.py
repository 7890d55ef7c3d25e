"""Line-block sharding of the `matchy match` hot path across the GPUs of one node.

The unit of work is a newline-terminated line block: no candidate class admits a line feed, so blocks
are independent, the database is replicated, and there is NO data-path collective. `torch.distributed` (RCCL on the
GPU box, gloo in the CPU tests) is used for exactly two things: the barrier around the timed region and the reduction
of the per-rank scalars (max time, summed bytes / lines / hits) that rank 0 reports.

Reference: the reference shards the same way on CPU — newline-aligned chunks handed to independent workers
(`crates/matchy/src/processing/parallel.rs:107-155, 411-439`), stats summed afterwards (`:478-520`).
"""
from dataclasses import dataclass


@dataclass
class Block:
    first_line: int
    n_lines: int


def block_for_rank(rank: int, world: int, lines_per_gpu: int) -> Block:
    """Weak scaling: every rank owns `lines_per_gpu` consecutive lines of the (virtual) global log."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return Block(first_line=rank * lines_per_gpu, n_lines=lines_per_gpu)


def strong_block_for_rank(rank: int, world: int, total_lines: int) -> Block:
    """Strong scaling: ONE job of `total_lines` lines cut into `world` contiguous line ranges whose sizes differ by at most
    one line (the generator is line-indexed, so every cut falls on a newline: the split_at_newlines rule for a real file)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(total_lines, world)
    first = rank * base + min(rank, rem)
    return Block(first_line=first, n_lines=base + (1 if rank < rem else 0))


def split_at_newlines(data: bytes, parts: int):
    """Strong-scaling helper for a real file: cut `data` into `parts` contiguous ranges that end on '\\n'
    (the last range takes the unterminated tail). Returns [(start, end)]; empty ranges are possible."""
    n = len(data)
    cuts = [0]
    for k in range(1, parts):
        target = max(cuts[-1], n * k // parts)
        nl = data.find(b"\n", target)
        cuts.append(n if nl < 0 else nl + 1)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(parts)]


def barrier(dist, world: int, sync=None):
    """Barrier + device synchronise on both sides of the timed region (bench.py contract)."""
    if sync is not None:
        sync()
    if world > 1:
        dist.barrier()
    if sync is not None:
        sync()


def aggregate(dist, world: int, device, elapsed_s: float, nbytes: int, lines: int, hits: int, candidates: int):
    """MAX of the per-rank elapsed time, SUM of the per-rank work. Returns a dict of Python scalars on every rank."""
    if world == 1:
        return {"elapsed_s": float(elapsed_s), "bytes": int(nbytes), "lines": int(lines), "hits": int(hits), "candidates": int(candidates)}
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([nbytes, lines, hits, candidates], dtype=torch.int64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    s = s.tolist()
    return {"elapsed_s": float(t.item()), "bytes": int(s[0]), "lines": int(s[1]), "hits": int(s[2]), "candidates": int(s[3])}
