"""Multi-GPU sharding of the scan (one process per GPU, `torch.distributed`; backend "nccl" is
RCCL on ROCm, "gloo" on CPU for tests).

The reference has no distributed path; records are independent (all per-record state is
re-initialised: src/GenomeMiner.jl:42,57, src/OmnGenomeMiner.jl:59,66,73-78), so the records of a
genome are split into contiguous, base-balanced shards, every rank scans its shard on its own
GPU with no data-path collective, and ONE variable-length gather brings the hit records to rank
0, which restores the reference's emission order and `genome_pos` bookkeeping.
"""
from __future__ import annotations

from typing import Optional, List, Sequence, Tuple

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

    The exchange is split in two so that it overlaps the next scan: `start` only ENQUEUES (pinned ->
    device copy, the collective, device -> pinned copy on rank 0; on a stream of its own, the host does
    not wait) and returns a slot; `finish(slot)` waits for that slot and merges.  A step loop calls
    `start(i)` and then `finish(i-1)`: the collective of step i runs while the library scans step i+1.
    Two slots are used alternately.  `gather` = `finish(start(...))`.
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
        self.on_gpu = on_gpu
        self.use_into_tensor = dist.get_backend(group) == "nccl"   # RCCL: straight into the stacked buffer
        self.stream = torch.cuda.Stream(device=self.dev) if on_gpu else None
        self.slots = []
        for _ in range(2):
            send_host = torch.zeros((rows, 8), dtype=torch.int64, pin_memory=on_gpu)
            recv_all = torch.zeros((self.world, rows, 8), dtype=torch.int64, device=self.dev)
            self.slots.append({
                "send_host": send_host, "send_np": send_host.numpy(),
                "send_dev": torch.zeros((rows, 8), dtype=torch.int64, device=self.dev) if on_gpu else send_host,
                "recv_all": recv_all, "recv_list": list(recv_all.unbind(0)),
                "recv_host": torch.zeros((self.world, rows, 8), dtype=torch.int64, pin_memory=True) if on_gpu else recv_all,
                # every rank reads the header rows back (64 B per rank): a capacity overflow anywhere is raised
                # on ALL ranks after the collective, never before it (a one-sided raise would leave the others
                # blocked in the next collective)
                "hdr_host": torch.zeros((self.world, 8), dtype=torch.int64, pin_memory=True) if on_gpu else None,
                "done": None,       # GPU: event after the last queued operation of the slot; CPU: the collective's work handle
                "busy": False,
            })
        self.turn = 0

    def start(self, hits: np.ndarray, contig_begin: int, genome_pos_local_advance: int) -> int:
        """Queue the exchange of this rank's `hits` (structured array of kgma_hit records); returns the slot."""
        n = int(hits.shape[0])
        n_send = min(n, self.cap)        # an overflow is reported through the header and raised by finish() on every rank
        t = self.turn
        self.turn ^= 1
        sl = self.slots[t]
        if sl["busy"]:
            raise RuntimeError("HitGatherer: both slots are in flight (call finish() for the older one first)")
        if self.on_gpu and sl["done"] is not None:
            sl["done"].synchronize()                # the slot's previous use has left its buffers (normally long done)
        buf = sl["send_np"]
        buf[0, 0] = n
        buf[0, 1] = int(genome_pos_local_advance)
        buf[0, 2] = int(contig_begin)
        if n_send:
            buf[1:1 + n_send] = np.ascontiguousarray(hits[:n_send]).view(np.int64).reshape(n_send, 8)
        if self.on_gpu:
            with self.torch.cuda.stream(self.stream):
                sl["send_dev"].copy_(sl["send_host"], non_blocking=True)
                if self.use_into_tensor:
                    work = self.dist.all_gather_into_tensor(sl["recv_all"], sl["send_dev"], group=self.group, async_op=True)
                else:
                    work = self.dist.all_gather(sl["recv_list"], sl["send_dev"], group=self.group, async_op=True)
                work.wait()                         # RCCL: orders self.stream after the collective, the host goes on
                if self.rank == 0:
                    sl["recv_host"].copy_(sl["recv_all"], non_blocking=True)
                sl["hdr_host"].copy_(sl["recv_all"][:, 0, :], non_blocking=True)
                ev = self.torch.cuda.Event()
                ev.record(self.stream)
                sl["done"] = ev
        else:
            sl["done"] = self.dist.all_gather(sl["recv_list"], sl["send_dev"], group=self.group, async_op=True)
        sl["busy"] = True
        return t

    def finish(self, slot: int):
        """Wait for the exchange queued in `slot`; returns the merged array on rank 0 (record indices and
        genome_pos made global), None elsewhere."""
        sl = self.slots[slot]
        if not sl["busy"]:
            raise RuntimeError("HitGatherer.finish: nothing in flight in this slot")
        sl["busy"] = False
        if self.on_gpu:
            sl["done"].synchronize()
            counts = sl["hdr_host"].numpy()[:, 0]
        else:
            sl["done"].wait()
            sl["done"] = None
            counts = sl["recv_all"].numpy()[:, 0, 0]
        over = [(r, int(c)) for r, c in enumerate(counts) if int(c) > self.cap]
        if over:                                    # every rank sees the same headers: all raise together
            raise RuntimeError("HitGatherer capacity %d exceeded: %s; construct it with a larger capacity"
                               % (self.cap, ", ".join("rank %d has %d hits" % rc for rc in over)))
        if self.rank != 0:
            return None
        blocks = sl["recv_host"].numpy()
        parts = []
        gp_off = 0
        for r in range(self.world):
            cnt, adv, begin = int(blocks[r, 0, 0]), int(blocks[r, 0, 1]), int(blocks[r, 0, 2])
            rec = blocks[r, 1:1 + cnt].copy().reshape(-1).view(HIT_RECORD_DTYPE)
            rec["contig"] += begin              # shard-local record index -> genome record index
            rec["genome_pos"] += gp_off         # genome_pos continues across shards
            parts.append(rec)
            gp_off += adv
        return np.concatenate(parts) if parts else np.zeros(0, dtype=HIT_RECORD_DTYPE)

    def gather(self, hits: np.ndarray, contig_begin: int, genome_pos_local_advance: int):
        """Blocking form: `finish(start(...))`."""
        return self.finish(self.start(hits, contig_begin, genome_pos_local_advance))


# ------------------------------------------------------------------------------------------------
# Sharding INSIDE records (balance by bases, not by record).  Windows are
# independent in exact arithmetic, so a record's windows are cut into slices; consecutive slices share
# ONE tested window, so every dip that leaves a slice has its exit value, and a dip that reaches the
# end of a slice continues in the dip that starts the next one.  Ranks ship dips (not hits); rank 0
# joins them and runs the reference's hit state machine once over whole records (kgma_replay_dips).
# ------------------------------------------------------------------------------------------------
def record_windows(L: int, mode_single: bool, windowsizes: Sequence[int], k: int) -> int:
    """Windows a record contributes (GenomeMiner.jl:37-39,60; OmnGenomeMiner.jl:89)."""
    if mode_single:
        W = int(windowsizes[0])
        return L - W + 1 if L >= W else 0
    n_iter = L - max(int(w) for w in windowsizes) - k + 2
    return n_iter + 1 if n_iter >= 1 else 0


def plan_slices(lengths: Sequence[int], world: int, mode_single: bool, windowsizes: Sequence[int], k: int,
                min_windows: int = 4096) -> List[List[Tuple[int, int, int]]]:
    """Per rank: slices (record, u, v) = windows u..v (1-based window starts) of that record.  A slice's
    first window is never tested: it is the record's first window (u = 1) or the previous slice's last
    but one (u = v_prev - 1, so window v_prev is tested by both slices).  Cuts fall every total/world
    windows; a slice takes at least `min_windows` new windows (>= 2)."""
    min_windows = max(int(min_windows), 2)
    nwin = [record_windows(int(L), mode_single, windowsizes, k) for L in lengths]
    total = sum(nwin)
    per = max((total + world - 1) // max(world, 1), 1)
    out: List[List[Tuple[int, int, int]]] = [[] for _ in range(world)]
    done = 0                                   # windows handed out so far, over all records
    for c, n in enumerate(nwin):
        if n == 0:
            continue                           # skipped record: nobody scans it, the replay still counts its length
        u, covered = 1, 0
        while True:
            rank = min(done // per, world - 1)
            room = (rank + 1) * per - done
            remaining = n - covered
            take = remaining if (rank == world - 1 or remaining <= room + min_windows) else max(room, min_windows)
            take = min(take, remaining)
            v = covered + take
            out[rank].append((c, u, v))
            done += take
            covered += take
            if covered >= n:
                break
            u = v - 1
    return out


def slice_bases(u: int, v: int, L: int, mode_single: bool, windowsizes: Sequence[int], k: int) -> Tuple[int, int]:
    """0-based [begin, end) residues a slice needs so that its local scan evaluates exactly windows u..v."""
    if mode_single:
        n = (v - u) + int(windowsizes[0])
    else:
        n = (v - u + 1) + max(int(w) for w in windowsizes) + k - 3
    return u - 1, min(u - 1 + n, L)


class RecordSource:
    """The records of a sharded scan as a rank sees them: every rank knows all record LENGTHS, but only reads the residues
    of its own slices through `fetch(record, begin, end)` (0-based half-open).  A plain list of bytes is wrapped as is."""

    def __init__(self, lengths: Sequence[int], fetch):
        self.lengths = [int(x) for x in lengths]
        self.fetch = fetch

    @staticmethod
    def of(records) -> "RecordSource":
        if isinstance(records, RecordSource):
            return records
        recs = list(records)
        return RecordSource([len(r) for r in recs], lambda c, b, e: recs[c][b:e])


def local_scan(ctx, records, my_slices: Sequence[Tuple[int, int, int]], mode: int, flags: int = 0, keep_genome: bool = False) -> dict:
    """One rank's part: scan the slices, decide local ties, return dips (and the guard-band windows) in whole-record
    coordinates.  keep_genome: the slices' device genome stays alive in payload["_genome"] (the caller frees it): the chain
    replay may come back for it (serve_chain_request)."""
    from . import _lib
    src = RecordSource.of(records)
    mode_single = mode == _lib.MODE_SINGLE
    ws = [ctx.ws[0]] if mode_single else list(ctx.ws)
    m = 1 if mode_single else len(ctx.ws)
    pieces = []
    for (c, u, v) in my_slices:
        b, e = slice_bases(u, v, src.lengths[c], mode_single, ws, ctx.k)
        pieces.append(src.fetch(c, b, e))
    payload = dict(slices=list(my_slices), dips=np.zeros(0, dtype=_lib.DIP_DTYPE), last_min=np.zeros(0, dtype=np.int64), first_D={},
                   att=np.zeros((0, 3), dtype=np.int64))
    if not pieces:
        return payload
    g = ctx.genome_from_host(pieces)
    try:
        ctx.scan_device(g, mode, flags & ~_lib.F_CHAIN_REPLAY)
        if not (flags & _lib.F_NO_TIE_RESOLVE):
            ctx.resolve_ties_local(g)
        dips = ctx.dips_array().copy()
        last_min = ctx.dip_last_min().copy()
        fw = [ctx.first_window(j + 1) for j in range(m)]
        att = ctx.att().copy()
    except Exception:
        g.free()
        raise
    if keep_genome:
        payload["_genome"] = g
    else:
        g.free()
    off = np.array([u - 1 for (_, u, _) in my_slices], dtype=np.int64)
    rec = np.array([c for (c, _, _) in my_slices], dtype=np.int32)
    if att.shape[0]:
        loc = att[:, 0]
        att[:, 2] += off[loc]
        att[:, 0] = rec[loc]
    payload["att"] = att
    if dips.size:
        loc = dips["contig"].astype(np.int64)
        o = off[loc]
        for f in ("start", "end", "argmin"):
            dips[f] += o
        dips["exit_pos"] = np.where(dips["exit_pos"] != 0, dips["exit_pos"] + o, 0)
        last_min = last_min + o
        dips["reserved"] = loc.astype(np.uint32)                     # slice order inside the rank (ties in sorting)
        dips["contig"] = rec[loc]
    payload["dips"], payload["last_min"] = dips, last_min
    for i, (c, u, _) in enumerate(my_slices):
        if u == 1:
            payload["first_D"][int(c)] = [int(fw[j][i]) for j in range(m)]
    return payload


# ---- the ranks' dips travel as ONE fixed-layout int64 tensor per rank (RCCL / gloo all_gather), like the hit records ----
_PAYLOAD_HEADER = 8       # n_dips, n_first, status, error class, error record, error position, rank, n_att


def encode_payload(payload: dict, m: int) -> np.ndarray:
    """[header | n_dips x 9 (the 64-byte kgma_dip as 8 words + the last window attaining the minimum) | n_first x (1 + m)]."""
    from . import _lib
    dips = np.ascontiguousarray(payload["dips"], dtype=_lib.DIP_DTYPE)
    nd = int(dips.shape[0])
    first = sorted(payload["first_D"].items())
    att = np.asarray(payload.get("att", np.zeros((0, 3), dtype=np.int64)), dtype=np.int64).reshape(-1, 3)
    out = np.zeros(_PAYLOAD_HEADER + nd * 9 + len(first) * (1 + m) + att.size, dtype=np.int64)
    out[0], out[1], out[7] = nd, len(first), att.shape[0]
    err = payload.get("error")
    if err:
        out[2:7] = [int(err["status"]), int(err["kind"]), int(err["record"]), int(err["position"]), int(err["rank"])]
    if nd:
        body = out[_PAYLOAD_HEADER:_PAYLOAD_HEADER + nd * 9].reshape(nd, 9)
        body[:, :8] = dips.view(np.int64).reshape(nd, 8)
        body[:, 8] = np.asarray(payload["last_min"], dtype=np.int64)
    p = _PAYLOAD_HEADER + nd * 9
    for c, vals in first:
        out[p] = c
        out[p + 1:p + 1 + m] = vals
        p += 1 + m
    out[p:p + att.size] = att.reshape(-1)
    return out


def decode_payload(buf: np.ndarray, m: int) -> dict:
    from . import _lib
    nd, nf = int(buf[0]), int(buf[1])
    body = buf[_PAYLOAD_HEADER:_PAYLOAD_HEADER + nd * 9].reshape(nd, 9)
    dips = np.ascontiguousarray(body[:, :8]).reshape(-1).view(_lib.DIP_DTYPE).copy() if nd else np.zeros(0, dtype=_lib.DIP_DTYPE)
    last_min = body[:, 8].copy() if nd else np.zeros(0, dtype=np.int64)
    first_D = {}
    p = _PAYLOAD_HEADER + nd * 9
    for _ in range(nf):
        first_D[int(buf[p])] = [int(x) for x in buf[p + 1:p + 1 + m]]
        p += 1 + m
    na = int(buf[7])
    att = buf[p:p + 3 * na].reshape(na, 3).copy()
    err = None
    if int(buf[2]) != 0:
        err = dict(status=int(buf[2]), kind=int(buf[3]), record=int(buf[4]), position=int(buf[5]), rank=int(buf[6]))
    return dict(dips=dips, last_min=last_min, first_D=first_D, error=err, att=att)


def gather_payloads(payload: dict, m: int, device=None, group=None) -> List[dict]:
    """Every rank's payload on every rank: an all_gather of the sizes, then ONE padded all_gather of the int64 blocks
    (device tensors with RCCL, host tensors with gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    enc = encode_payload(payload, m)
    size = torch.tensor([enc.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    mx = max(int(x) for x in sizes)
    buf = torch.zeros(mx, dtype=torch.int64, device=dev)
    buf[:enc.size] = torch.from_numpy(enc).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    return [decode_payload(bufs[r][:int(sizes[r])].cpu().numpy(), m) for r in range(world)]


def _gather_int64(enc: np.ndarray, device=None, group=None) -> List[np.ndarray]:
    """Every rank's int64 block on every rank (sizes first, then one padded all_gather)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    size = torch.tensor([enc.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    mx = max(max(int(x) for x in sizes), 1)
    buf = torch.zeros(mx, dtype=torch.int64, device=dev)
    if enc.size:
        buf[:enc.size] = torch.from_numpy(np.ascontiguousarray(enc)).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    return [bufs[r][:int(sizes[r])].cpu().numpy() for r in range(world)]


def _bcast_int64(enc: Optional[np.ndarray], src: int = 0, device=None, group=None) -> np.ndarray:
    """Rank `src`'s int64 block on every rank (its size first)."""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    size = torch.tensor([0 if enc is None else int(enc.size)], dtype=torch.int64, device=dev)
    dist.broadcast(size, src, group=group)
    n = int(size)
    buf = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
    if enc is not None and n:
        buf[:n] = torch.from_numpy(np.ascontiguousarray(enc)).to(dev)
    dist.broadcast(buf, src, group=group)
    return buf[:n].cpu().numpy()


# ---- chain replay of a sharded scan (KGMA_F_CHAIN_REPLAY): the running Float64 value of a record is produced piece by piece
#      where the residues are -- every rank runs the chain kernel over its slices of the record (kgma_chain_export; a slice
#      stops one transition before the next begins), the pieces travel to rank 0 as int64 blocks, and rank 0 walks them in
#      order, each piece starting on the value the previous one ended on (kgma_host_chain_walk) ------------------------
def encode_chain_request(pairs) -> np.ndarray:
    out = [len(pairs)]
    for c, j, iv in pairs:
        out += [c, j, len(iv)]
        for a, b in iv:
            out += [a, b]
    return np.asarray(out, dtype=np.int64)


def decode_chain_request(buf: np.ndarray):
    pairs, p = [], 1
    for _ in range(int(buf[0])):
        c, j, n = int(buf[p]), int(buf[p + 1]), int(buf[p + 2])
        p += 3
        iv = [(int(buf[p + 2 * i]), int(buf[p + 2 * i + 1])) for i in range(n)]
        p += 2 * n
        pairs.append((c, j, iv))
    return pairs


def serve_chain_request(ctx, genome, my_slices, nwin_of, pairs) -> np.ndarray:
    """This rank's pieces of the requested chains: for every pair and every slice of mine of that record, the chain kernel's
    output over the slice's windows (local coordinates), as one int64 block:
    [n_pieces | per piece: record, kfv, u, local_last, n_streams, n_chunks, n_pool, first (bits), status, win0.., n_valid..,
     chunk_base.., D0.., chunks (2 words each), pool (2 words each)]."""
    from . import _lib
    pieces = []
    for c, j, iv in pairs:
        last_needed = iv[-1][1]
        for i, (cc, u, v) in enumerate(my_slices):
            if cc != c or genome is None:
                continue
            v_eff = v if v >= nwin_of[c] else v - 1              # a slice stops one transition before the next one begins
            g_last = min(v_eff, last_needed)
            if g_last <= u and u != 1:
                continue
            local_last = max(g_last - u + 1, 2)
            if v - u + 1 < 2:
                continue
            hot = [(max(a, u + 1) - u + 1, min(b, g_last) - u + 1) for a, b in iv if min(b, g_last) >= max(a, u + 1)]
            if not hot or hot[-1][1] < local_last:
                hot.append((local_last, local_last))                # the value the next piece starts on
            head = [c, j, u, local_last, 0, 0, 0, 0, 0]
            try:
                ex = ctx.chain_export(genome, i, j, local_last, hot)
                head[4:8] = [ex["win0"].size, ex["chunks"].size, ex["pool"].size, int(np.float64(ex["first"]).view(np.int64))]
                body = [ex["win0"].astype(np.int64), ex["n_valid"].astype(np.int64), ex["chunk_base"].astype(np.int64), ex["D0"].astype(np.int64),
                        np.ascontiguousarray(ex["chunks"]).view(np.int64).reshape(-1), np.ascontiguousarray(ex["pool"]).view(np.int64).reshape(-1)]
            except Exception as e:                                  # (any failure travels as the piece's status: the gather must complete on every rank)
                head[4:8] = [0, 0, 0, 0]
                head[8] = int(getattr(e, "status", 0)) or 1
                body = []
            pieces.append(np.concatenate([np.asarray(head, dtype=np.int64)] + body))
    return np.concatenate([np.asarray([len(pieces)], dtype=np.int64)] + pieces) if pieces else np.asarray([0], dtype=np.int64)


def decode_chain_pieces(buf: np.ndarray) -> List[dict]:
    from . import _lib
    out, p = [], 1
    for _ in range(int(buf[0])):
        c, j, u, local_last, ns, nc, npool, fbits, status = (int(x) for x in buf[p:p + 9])
        p += 9
        d = dict(record=c, kfv=j, u=u, local_last=local_last, status=status)
        if status == 0:
            d["first"] = float(np.int64(fbits).view(np.float64))
            d["win0"], d["n_valid"], d["chunk_base"], d["D0"] = (buf[p + t * ns:p + (t + 1) * ns].copy() for t in range(4))
            p += 4 * ns
            d["chunks"] = buf[p:p + 2 * nc].copy().view(_lib.CHAIN_CHUNK_DTYPE)
            p += 2 * nc
            d["pool"] = buf[p:p + 2 * npool].copy().view(_lib.CHAIN_CHUNK_DTYPE)
            p += 2 * npool
        out.append(d)
    return out


def walk_chain_pieces(ctx, pairs, pieces: List[dict], ws: Sequence[int]) -> List[np.ndarray]:
    """Rank 0: the values at every pair's wanted windows, from the pieces of all ranks."""
    from . import _lib
    out = []
    for c, j, iv in pairs:
        mine = sorted((d for d in pieces if d["record"] == c and d["kfv"] == j), key=lambda d: d["u"])
        if any(d["status"] for d in mine):
            raise _lib.KgmaError(next(d["status"] for d in mine if d["status"]), f"a rank could not run the chain kernel for record {c} KFV {j}")
        scale, nk = ctx.kfv_scale(j), int(ws[j - 1]) - ctx.k + 1
        vals = {}
        v, reached = None, 0                                       # value at global window `reached`
        for d in mine:
            u, ll = d["u"], d["local_last"]
            if u == 1:
                v, reached = d["first"], 1
            if v is None or reached != u:
                raise _lib.KgmaError(_lib.KGMA_E_STATE, f"chain pieces of record {c} KFV {j} do not join at window {u} (reached {reached})")
            g_last = u + ll - 1
            want = [(max(a, u + 1 if u > 1 else 1), min(b, g_last)) for a, b in iv]
            want = [(a - u + 1, b - u + 1) for a, b in want if b >= a]
            if not want or want[-1][1] < ll:
                want.append((ll, ll))                              # the value the next piece starts on
            got, _ = _lib.host_chain_walk(v, scale, nk, d["win0"], d["n_valid"], d["chunk_base"], d["D0"], d["chunks"], d["pool"], want)
            q = 0
            for a, b in want:
                for w in range(a, b + 1):
                    vals[w + u - 1] = got[q]
                    q += 1
            v, reached = vals[g_last], g_last
        res = np.asarray([vals[w] for a, b in iv for w in range(a, b + 1)], dtype=np.float64)
        out.append(res)
    return out


def merge_payloads(payloads: Sequence[dict], n_records: int, m: int, float_kfv: Optional[Sequence[bool]] = None):
    """Join the ranks' dips: sort by (record, KFV, start), merge a dip that reaches the end of its slice
    with the dip that starts the next slice at the shared window.  Returns (dips, last_min, first_D).

    `float_kfv[j]` (Context.kfv_is_float): KFV j + 1 is a general Float64 vector.  Its D values are anchored per stream, so two
    ranks report the shared window (and any two windows whose distances differ by rounding noise) a few units apart: minima within
    the library's near-tie tolerance max(|a|, |b|) * 2^-30 + 2 (kgma_kfv_is_float, kgma.h) count as tied, as they do between
    the streams of one rank."""
    from . import _lib
    TIE, RESOLVED = _lib.HIT_TIE, _lib.HIT_TIE_RESOLVED
    dips = np.concatenate([p["dips"] for p in payloads]) if payloads else np.zeros(0, dtype=_lib.DIP_DTYPE)
    last_min = np.concatenate([p["last_min"] for p in payloads]) if payloads else np.zeros(0, dtype=np.int64)
    first_D = np.full((m, n_records), -1, dtype=np.int64)
    for p in payloads:
        for c, vals in p["first_D"].items():
            first_D[:, c] = vals
    if dips.size == 0:
        return dips, last_min, first_D
    order = np.lexsort((dips["end"], dips["start"], dips["kfv"], dips["contig"]))
    dips, last_min = dips[order], last_min[order]
    out, out_last = [], []
    cur, cur_last = None, 0
    for d, lm in zip(dips, last_min):
        d = d.copy()
        if (cur is not None and cur["contig"] == d["contig"] and cur["kfv"] == d["kfv"] and cur["exit_pos"] == 0
                and cur["end"] == d["start"]):
            # the shared window is under the threshold in both slices: one dip
            a, b = int(cur["D_min"]), int(d["D_min"])
            tol = 0
            if float_kfv is not None and float_kfv[int(d["kfv"]) - 1]:
                tol = int(max(abs(a), abs(b)) * 2.0 ** -30) + 2
            if b < a - tol:
                cur["D_min"], cur["argmin"], cur_last = d["D_min"], d["argmin"], lm
                cur["flags"] = (cur["flags"] & ~np.uint32(TIE | RESOLVED)) | (d["flags"] & np.uint32(TIE | RESOLVED))
            elif b <= a + tol:
                single = cur["argmin"] == cur_last == d["argmin"] == lm           # the minimum IS the shared window
                if b < a:
                    cur["D_min"], cur["argmin"] = d["D_min"], d["argmin"]
                cur_last = lm
                if not single:
                    cur["flags"] = (cur["flags"] & ~np.uint32(RESOLVED)) | np.uint32(TIE)   # tie across a slice boundary
                cur["flags"] |= d["flags"] & np.uint32(TIE)
            cur["flags"] |= d["flags"] & np.uint32(_lib.HIT_AT_THRESHOLD)
            cur["end"], cur["exit_pos"], cur["D_exit"] = d["end"], d["exit_pos"], d["D_exit"]
            continue
        if cur is not None:
            out.append(cur); out_last.append(cur_last)
        cur, cur_last = d, lm
    if cur is not None:
        out.append(cur); out_last.append(cur_last)
    merged = np.array(out, dtype=_lib.DIP_DTYPE)
    merged["reserved"] = 0
    return merged, np.array(out_last, dtype=np.int64), first_D


_ERR_KINDS = {"KgmaError": 0, "BadBaseError": 1, "RecordBoundsError": 2}


def _error_entry(e, plan_rank, src, mode_single, ws, k, rank) -> dict:
    """A rank-local library error in genome coordinates: the slice-local record index / position of the message are
    mapped back through the rank's slices."""
    import re
    kind = _ERR_KINDS.get(type(e).__name__, 0)
    rec, pos = (plan_rank[0][0] if plan_rank else 0), 0
    mt = re.search(r"record (\d+)(?: position (\d+))?", e.message or "")
    if mt and int(mt.group(1)) < len(plan_rank):
        c, u, v = plan_rank[int(mt.group(1))]
        b, _ = slice_bases(u, v, src.lengths[c], mode_single, ws, k)
        rec = c
        pos = b + int(mt.group(2)) if mt.group(2) else 0
    return dict(status=int(e.status), kind=kind, record=int(rec), position=int(pos), rank=int(rank))


def scan_sharded(ctx, records, mode: int, buff: int = 50, genome_pos: int = 0, flags: int = 0,
                 align=None, group=None, min_windows: int = 4096, device=None):
    """findGenes-style scan of `records` sharded over all ranks INSIDE records.  `records`: a list of bytes (every rank
    holds everything) or a RecordSource (every rank knows the lengths and reads only its slices).  The ranks' dips are
    gathered as one int64 tensor per rank (`device` = the rank's GPU for RCCL, None for gloo); rank 0 joins them, runs the
    hit state machine over whole records and returns the hits (list of dicts in the reference's order), the others [].
    Rank 0 reads the few residues it needs to decide ties between a dip and the stale running minimum through the same
    source (kgma_set_residue_source)."""
    import torch.distributed as dist
    from . import _lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    src = RecordSource.of(records)
    mode_single = mode == _lib.MODE_SINGLE
    ws = [ctx.ws[0]] if mode_single else list(ctx.ws)
    m = 1 if mode_single else len(ctx.ws)
    lengths = src.lengths
    plan = plan_slices(lengths, world, mode_single, ws, ctx.k, min_windows)
    # a rank-local failure (a residue outside A/C/G/T/N in this rank's slices, a record shorter than k-1, a
    # record-buffer overflow) must not leave the other ranks blocked in the collective: it travels in the header of the
    # rank's block, the collective completes, and EVERY rank raises the first error in record order
    chain = bool(flags & _lib.F_CHAIN_REPLAY) and not (flags & _lib.F_NO_TIE_RESOLVE)
    genome = None
    try:
        payload = local_scan(ctx, src, plan[rank], mode, flags, keep_genome=chain)
        genome = payload.pop("_genome", None)
        payload["error"] = None
    except _lib.KgmaError as e:
        payload = dict(slices=list(plan[rank]), dips=np.zeros(0, dtype=_lib.DIP_DTYPE), last_min=np.zeros(0, dtype=np.int64),
                       first_D={}, att=np.zeros((0, 3), dtype=np.int64), error=_error_entry(e, plan[rank], src, mode_single, ws, ctx.k, rank))
    try:
        gathered = gather_payloads(payload, m, device=device, group=group)
        errors = sorted((p["error"] for p in gathered if p.get("error")), key=lambda t: (t["record"], t["position"], t["rank"]))
        if errors:
            e = errors[0]
            if e["kind"] == 1:
                raise _lib.BadBaseError(e["status"], f"record {e['record']} position {e['position']}: residue is not one of A/C/G/T/N "
                                                     f"(KeyError, Consts.jl:22-28) [found by rank {e['rank']}]")
            if e["kind"] == 2:
                raise _lib.RecordBoundsError(e["status"], f"record {e['record']} has fewer than k-1 residues (BoundsError, "
                                                          f"OmnGenomeMiner.jl:84-86) [found by rank {e['rank']}]")
            raise _lib.KgmaError(e["status"], f"rank {e['rank']} failed while scanning its slices of record {e['record']}")
        nwin_of = [record_windows(int(L), mode_single, ws, ctx.k) for L in lengths]
        if rank != 0:
            # serve rank 0's chain requests (none, or one per replay) until it says it is done
            while chain:
                req = _bcast_int64(None, 0, device=device, group=group)
                if req.size == 0:
                    break
                _gather_int64(serve_chain_request(ctx, genome, plan[rank], nwin_of, decode_chain_request(req)), device=device, group=group)
            return []
        # From here on the other ranks sit in their serve loop (chain mode): whatever happens on rank 0 -- also before the replay
        # starts -- the release broadcast in the finally below must go out, or they block in the collective for ever.
        try:
            dips, last_min, first_D = merge_payloads(gathered, len(lengths), m, [ctx.kfv_is_float(j + 1) for j in range(m)])
            ctx.set_residue_source(lambda c, pos, n: src.fetch(c, pos - 1, pos - 1 + n))
            if chain:
                att = np.concatenate([p["att"] for p in gathered]) if gathered else np.zeros((0, 3), dtype=np.int64)
                ctx.set_att(np.unique(att, axis=0) if att.shape[0] else att)     # (a window shared by two slices is reported by both)

                def source(pairs):
                    # (the broadcast and the gather form one exchange: nothing that can raise sits between them --
                    #  serve_chain_request reports its failures as piece statuses)
                    _bcast_int64(encode_chain_request(pairs), 0, device=device, group=group)
                    blocks = _gather_int64(serve_chain_request(ctx, genome, plan[0], nwin_of, pairs), device=device, group=group)
                    pieces = [d for b in blocks for d in decode_chain_pieces(b)]
                    return walk_chain_pieces(ctx, pairs, pieces, ws if not mode_single else [ctx.ws[0]])
                ctx.set_chain_source(source)
            ctx.replay_dips(mode, buff, genome_pos, flags, lengths, first_D, dips, last_min, align)
        finally:
            ctx.set_residue_source(None)
            if chain:
                ctx.set_chain_source(None)
                _bcast_int64(np.zeros(0, dtype=np.int64), 0, device=device, group=group)     # releases the other ranks
        return ctx.hits()
    finally:
        if genome is not None:
            genome.free()
