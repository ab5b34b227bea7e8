"""numpy restatement of what stream8_kernel<..., CHAIN> writes (kgma_device.h: ChainChunk), for the CPU tests of the
host half of the device chain: per-transition Float64 increments in the reference's operation order
(src/GenomeMiner.jl:70-72), chunk translations for both parities of the incoming value, raw increments where a window may
leave its binade or the chunk is hot.  Test infrastructure only."""
import numpy as np

from kmergma_amd import _lib

STEPS = _lib.chain_steps()
GUARD = 2.0 ** -29

_CODE = np.full(256, 3, dtype=np.int64)
for _ch, _v in ((b"A", 0), (b"C", 1), (b"G", 2), (b"a", 0), (b"c", 1), (b"g", 2)):
    _CODE[_ch[0]] = _v


def increments(seq: bytes, ref, S, N: int, k: int, W: int):
    """(first, inc[nwin-1], D[nwin]): the first window's Float64 distance (left-to-right sqeuclidean), the increment of
    every transition (0.0 where left == right) and every window's exact integer D."""
    code = _CODE[np.frombuffer(seq, dtype=np.uint8)]
    L = len(seq)
    nb = 4 ** k
    km = np.zeros(L - k + 1, dtype=np.int64)
    for i in range(k):
        km = km * 4 + code[i:L - k + 1 + i]
    nk = W - k + 1
    cnt = np.zeros(nb, dtype=np.int64)
    np.add.at(cnt, km[:nk], 1)
    ref = np.asarray(ref, dtype=np.float64)
    S = np.asarray(S, dtype=np.int64)
    sq = 0.0
    for x in range(nb):
        d = ref[x] - float(cnt[x])
        sq += d * d
    SF = 1.0 / k
    first = (SF * 0.5) * sq
    nwin = L - W + 1
    inc = np.zeros(nwin - 1, dtype=np.float64)
    D = np.zeros(nwin, dtype=np.int64)
    D[0] = int(((S - N * cnt) ** 2).sum())
    cl_ = cnt.tolist()
    refl = ref.tolist()
    Sl = S.tolist()
    kml = km.tolist()
    Dw = int(D[0])
    for w in range(1, nwin):
        l, r = kml[w - 1], kml[w - 1 + nk]
        if l != r:
            cl, cr = cl_[l], cl_[r]
            t = float(1 + cr)
            t = t + refl[l]
            t = t - refl[r]
            t = t - float(cl)
            inc[w - 1] = SF * t
            Dw += 2 * N * ((Sl[l] - Sl[r]) - N * (cl - 1 - cr))
            cl_[l] = cl - 1
            cl_[r] = cr + 1
        D[w] = Dw
    return first, inc, D


def _binade(Dv: int, scale: float):
    if Dv <= 0:
        return None
    e = int(np.floor(np.log2(Dv / scale)))
    lo, hi = np.ldexp(scale, e) * (1 + GUARD), np.ldexp(scale, e + 1) * (1 - GUARD)
    if not (lo < Dv < hi):
        return None
    return e, lo, hi


def emulate(inc, D, nk: int, T: int, scale: float, hot_windows=(), last=None):
    """Streams of T transitions over windows 1..last, as the kernel lays them out: per chunk one record (its leading run
    of regular steps), and for a chunk with raw steps -- hot ones, or steps in which a window may leave the binade -- a list
    of entries in the pool (later runs and raw steps, in order).  Returns dict(win0, n_valid, chunk_base, D0, chunks, pool)."""
    nwin = len(D)
    last = nwin if last is None else last
    hot_windows = set(int(w) for w in hot_windows)
    win0s, nvs, cbases, D0s = [], [], [], []
    chunks = []
    pool = []                                        # units: (A0, info, raw) tuples or raw arrays of 64 doubles (32 units)
    pool_units = [0]

    def pool_raw(vals):
        pool.append(("r", vals)); pool_units[0] += 32
        return pool_units[0] - 32

    for win0 in range(1, last, T):
        n_valid = min(T + 1, last - win0 + 1)
        n_pos = n_valid + nk - 1
        n_blocks = (n_pos + 63) // 64
        win0s.append(win0); nvs.append(n_valid); cbases.append(len(chunks)); D0s.append(int(D[win0 - 1]))
        bin_ = None

        def lanes(b):
            p = np.arange(b * 64, (b + 1) * 64)
            q = p - nk + 1
            act = (q >= 1) & (q < n_valid)
            return act, win0 + q - 1                  # transition t leads from window t to t + 1

        for cb in range(0, n_blocks, STEPS):
            steps = min(STEPS, n_blocks - cb)
            acc = corr = dA = par = 0
            split = True
            run0 = 0
            detailed = False
            rec = None                                # the chunk's own record
            ents = []                                 # entries (in order) once detailed
            for step in range(steps):
                b = cb + step
                act, t = lanes(b)
                hot = any((int(x) + 1) in hot_windows for x in t[act])
                iv = np.where(act, inc[np.clip(t - 1, 0, len(inc) - 1)], 0.0)
                raw = hot
                if not raw and act.any():
                    if bin_ is None:
                        if b * 64 >= nk:
                            prev_q = min(max(b * 64 - nk, 0), n_valid - 1)      # window the step starts on (local)
                            bin_ = _binade(int(D[win0 - 1 + prev_q]), scale)
                        elif b * 64 + 63 >= nk - 1:
                            bin_ = _binade(int(D[win0 - 1]), scale)               # the step that completes the warm-up knows D0
                    ok = bin_ is not None
                    if ok:
                        e, lo, hi = bin_
                        Da = D[t[act]]                                            # D after each active transition (window t + 1)
                        ok = bool(np.all((Da > lo) & (Da < hi)))
                    raw = not ok
                if raw:
                    n = step - run0
                    if not detailed:
                        rec = (acc + corr, (dA + 1) | (n << 2) | _lib.CHAIN_DETAIL)
                        detailed = True
                    elif n > 0:
                        ents.append(("e", (acc + corr, (dA + 1) | (n << 2), 0)))
                    ents.append(("r", iv))
                    acc = corr = dA = par = 0; split = True; run0 = step + 1
                    bin_ = None
                    continue
                if not act.any():
                    continue
                neg = np.signbit(iv)
                x0 = np.where(neg, np.ldexp(1.0, e + 1) - 2 * np.ldexp(1.0, e - 52), np.ldexp(1.0, e))
                x1 = (x0.view(np.int64) | 1).view(np.float64)
                r0 = x0 + iv
                r1 = x1 + iv
                a = r0.view(np.int64) - x0.view(np.int64)
                delta = (r1.view(np.int64) - r0.view(np.int64)) - 1
                acc += int(a.sum())
                for u in range(64):
                    if delta[u] != 0:
                        c0 = int(delta[u]) if par else 0
                        if split:
                            dA = (0 if par else int(delta[u])) - c0
                            split = False
                        corr += c0
                        par = 0
                    else:
                        par ^= int(a[u]) & 1
            n = steps - run0
            if not detailed:
                chunks.append((acc + corr, (dA + 1) | (n << 2), 0))
            else:
                if n > 0:
                    ents.append(("e", (acc + corr, (dA + 1) | (n << 2), 0)))
                # the entry list is contiguous; the raw blocks follow anywhere in the pool
                base = pool_units[0]
                where = []
                for _ in ents:
                    pool.append(None); pool_units[0] += 1
                    where.append(len(pool) - 1)
                for (kind, val), pos in zip(ents, where):
                    pool[pos] = ("e", val) if kind == "e" else ("e", (0, 1 | (1 << 2) | _lib.CHAIN_RAW, pool_raw(val)))
                chunks.append((rec[0], rec[1], base))
    ch = np.zeros(len(chunks), dtype=_lib.CHAIN_CHUNK_DTYPE)
    for i, c3 in enumerate(chunks):
        ch[i] = c3
    pl = np.zeros(pool_units[0], dtype=_lib.CHAIN_CHUNK_DTYPE)
    u = 0
    for kind, val in pool:
        if kind == "e":
            pl[u] = val
            u += 1
        else:
            pl[u:u + 32].view(np.float64)[:] = val
            u += 32
    return dict(win0=np.array(win0s), n_valid=np.array(nvs), chunk_base=np.array(cbases), D0=np.array(D0s), chunks=ch, pool=pl)
