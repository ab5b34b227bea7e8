"""Multi-GPU sharding of the scan (one process per GPU, `torch.distributed`; backend "nccl" is
RCCL on ROCm, "gloo" on CPU for tests).

The reference has no distributed path; records are independent (all per-record state is
re-initialised: src/GenomeMiner.jl:42,57, src/OmnGenomeMiner.jl:59,66,73-78), so the records of a
genome are split into contiguous, base-balanced shards, every rank scans its shard on its own
GPU with no data-path collective, and ONE variable-length gather brings the hit records to rank
0, which restores the reference's emission order and `genome_pos` bookkeeping.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

HIT_FIELDS = ("contig", "kfv", "cmi", "lo", "hi", "genome_pos", "D", "flags")


def shard_contigs(lengths: Sequence[int], world_size: int) -> List[Tuple[int, int]]:
    """Contiguous [begin, end) record ranges per rank, balanced by bases (greedy on the prefix sum)."""
    n = len(lengths)
    pre = np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])
    total = int(pre[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        j = int(np.searchsorted(pre, target, side="left"))
        if j > 0 and j <= n and abs(pre[j - 1] - target) <= abs(pre[min(j, n)] - target):
            j -= 1
        cuts.append(min(max(j, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def genome_pos_advance(lengths: Sequence[int], mode_single: bool, windowsize: int) -> int:
    """Bases by which a shard advances `genome_pos`: the single engine skips records shorter than
    the window (GenomeMiner.jl:37-39,106); the cluster engine counts all (OmnGenomeMiner.jl:159)."""
    if mode_single:
        return int(sum(L for L in lengths if L >= windowsize))
    return int(sum(lengths))


def encode_hits(hits: Sequence[dict]) -> np.ndarray:
    out = np.zeros((len(hits), len(HIT_FIELDS)), dtype=np.int64)
    for i, h in enumerate(hits):
        for j, f in enumerate(HIT_FIELDS):
            out[i, j] = int(h[f])
    return out


def decode_hits(arr: np.ndarray, scale_of_kfv) -> List[dict]:
    hits = []
    for row in arr:
        h = {f: int(v) for f, v in zip(HIT_FIELDS, row)}
        h["dist"] = h["D"] / scale_of_kfv(h["kfv"])
        hits.append(h)
    return hits


def gather_hits(local_hits: Sequence[dict], contig_begin: int, genome_pos_local_advance: int, scale_of_kfv,
                device=None, group=None) -> List[dict]:
    """Gather every rank's hits on rank 0 (other ranks get []).

    `contig_begin`: index of the rank's first record in the whole genome; `genome_pos_local_advance`:
    see genome_pos_advance().  Two collectives: an all_gather of (count, advance) and one padded
    all_gather of the records (KB-MB payload, latency bound; no ring all-reduce anywhere).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    enc = encode_hits(local_hits)
    meta = torch.tensor([enc.shape[0], int(genome_pos_local_advance), int(contig_begin)], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    mx = max(max(counts), 1)
    buf = torch.zeros((mx, len(HIT_FIELDS)), dtype=torch.int64, device=dev)
    if enc.shape[0]:
        buf[:enc.shape[0]] = torch.from_numpy(enc).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    if rank != 0:
        return []
    out: List[dict] = []
    gp_off = 0
    for r in range(world):
        arr = bufs[r][:counts[r]].cpu().numpy().copy()
        if arr.size:
            arr[:, 0] += int(metas[r][2])       # shard-local record index -> genome record index
            arr[:, 5] += gp_off                 # genome_pos continues across shards
        out.extend(decode_hits(arr, scale_of_kfv))
        gp_off += int(metas[r][1])
    return out
