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


# 64-byte kgma_hit records travel as 8 int64 words each (same layout as _lib.HIT_DTYPE)
HIT_RECORD_DTYPE = np.dtype([("contig", "<i4"), ("kfv", "<i4"), ("cmi", "<i8"), ("lo", "<i8"), ("hi", "<i8"),
                             ("genome_pos", "<i8"), ("dist", "<f8"), ("D", "<i8"), ("flags", "<u4"), ("reserved", "<u4")])


class HitGatherer:
    """Per-step gather of kgma_hit records with ONE collective and preallocated buffers.

    Every rank contributes a fixed-capacity block `[header | cap records]` (header = count, genome_pos
    advance, first record index); one `all_gather` brings the blocks to every rank and only rank 0
    copies them to the host, fixes record indices / `genome_pos` and returns a structured array.  The
    raw 64-byte records are shipped as they come out of `kgma_get_hits`: no per-hit Python work.
    `gather_hits` above is the general (any count, list-of-dict) form of the same exchange.
    """

    def __init__(self, device=None, group=None, capacity: int = 256):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.cap = int(capacity)
        self.dev = device if device is not None else torch.device("cpu")
        on_gpu = self.dev.type != "cpu"
        rows = 1 + self.cap
        # two pinned send buffers used alternately: a rank that runs ahead must not overwrite a block whose
        # host->device copy is still queued behind the previous step's collective
        self.send_host = [torch.zeros((rows, 8), dtype=torch.int64, pin_memory=on_gpu) for _ in range(2)]
        self.send_np = [t.numpy() for t in self.send_host]
        self.send_dev = [torch.zeros((rows, 8), dtype=torch.int64, device=self.dev) for _ in range(2)] if on_gpu else self.send_host
        self.copied = [None, None]
        self.turn = 0
        self.recv_all = torch.zeros((self.world, rows, 8), dtype=torch.int64, device=self.dev)
        self.recv_list = list(self.recv_all.unbind(0))
        self.recv_host = self.recv_all if not on_gpu else torch.zeros((self.world, rows, 8), dtype=torch.int64, pin_memory=True)
        self.on_gpu = on_gpu
        self.use_into_tensor = dist.get_backend(group) == "nccl"   # RCCL: straight into the stacked buffer

    def gather(self, hits: np.ndarray, contig_begin: int, genome_pos_local_advance: int):
        """`hits`: structured array of the rank's kgma_hit records.  Returns the merged array on rank 0
        (record indices and genome_pos made global), None elsewhere."""
        n = int(hits.shape[0])
        if n > self.cap:
            raise RuntimeError(f"HitGatherer capacity {self.cap} < {n} hits: construct it with a larger capacity")
        t = self.turn
        self.turn ^= 1
        if self.copied[t] is not None:
            self.copied[t].synchronize()            # (normally long done)
        buf = self.send_np[t]
        buf[0, 0] = n
        buf[0, 1] = int(genome_pos_local_advance)
        buf[0, 2] = int(contig_begin)
        if n:
            buf[1:1 + n] = np.ascontiguousarray(hits).view(np.int64).reshape(n, 8)
        if self.on_gpu:
            self.send_dev[t].copy_(self.send_host[t], non_blocking=True)
            ev = self.torch.cuda.Event()
            ev.record(self.torch.cuda.current_stream(self.dev))
            self.copied[t] = ev
        if self.use_into_tensor:
            self.dist.all_gather_into_tensor(self.recv_all, self.send_dev[t], group=self.group)
        else:
            self.dist.all_gather(self.recv_list, self.send_dev[t], group=self.group)
        if self.rank != 0:
            return None
        if self.on_gpu:
            self.recv_host.copy_(self.recv_all, non_blocking=True)
            self.torch.cuda.current_stream(self.dev).synchronize()
        blocks = self.recv_host.numpy()
        parts = []
        gp_off = 0
        for r in range(self.world):
            cnt, adv, begin = int(blocks[r, 0, 0]), int(blocks[r, 0, 1]), int(blocks[r, 0, 2])
            if cnt > self.cap:
                raise RuntimeError(f"rank {r} reported {cnt} hits, capacity {self.cap}")
            rec = blocks[r, 1:1 + cnt].copy().reshape(-1).view(HIT_RECORD_DTYPE)
            rec["contig"] += begin              # shard-local record index -> genome record index
            rec["genome_pos"] += gp_off         # genome_pos continues across shards
            parts.append(rec)
            gp_off += adv
        return np.concatenate(parts) if parts else np.zeros(0, dtype=HIT_RECORD_DTYPE)
