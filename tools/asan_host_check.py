#!/usr/bin/env python3
"""The library's HOST-only entry points under AddressSanitizer / UBSan (the CPU build: GPU sanitizers are not available on the pool):
kgma_host_chain_values against the oracle, kgma_host_semiglobal_cigar incl. a too-small CIGAR buffer, kgma_host_chain_walk on hostile
chunk records (every step raw although few windows are wanted; wanted windows whose steps are NOT raw: the walk must end with a
status, not write past its buffer -- ADVICE r3).

usage:  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -shared \
            -o /tmp/libkgma_hostasan.so kmergma.jl_amd/csrc/kgma_chain.cpp kmergma.jl_amd/csrc/kgma_align_host.cpp -lpthread
        LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_host_check.py
"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]
L = C.CDLL("/tmp/libkgma_hostasan.so")
from tests import chain_emul
from oracle import oracle as orc
rng = np.random.default_rng(1)
B = np.frombuffer(b"ACGT", dtype=np.uint8)
i64, dbl = C.c_int64, C.c_double
def P(a, t): return a.ctypes.data_as(C.POINTER(t))
# 1. host chain values vs oracle
seq = B[rng.integers(0, 4, size=20000)].tobytes()
k, W = 6, 289
RV = rng.random(4 ** k)
lo = np.array([1, 500, 19000], dtype=np.int64); hi = np.array([3, 700, 19712], dtype=np.int64)
out = np.zeros(int((hi - lo + 1).sum())); n = i64(0)
rc = L.kgma_host_chain_values(seq, i64(len(seq)), P(RV, dbl), C.c_int32(k), i64(W), P(lo, i64), P(hi, i64), i64(3), P(out, dbl), i64(out.size), C.byref(n))
_, od = orc.single_scan([seq], RV, k, W, 1.0, 50, return_dists=True)
chain = np.concatenate([[orc.kmer_dist_kfv(seq[:W], RV, k)], od])
want = np.concatenate([chain[a - 1:b] for a, b in zip(lo, hi)])
assert rc == 0 and np.array_equal(out, want), rc
print("host chain values ok")
# 2. aligner
a = B[rng.integers(0, 4, size=300)].tobytes(); b = B[rng.integers(0, 4, size=520)].tobytes()
cig = C.create_string_buffer(4096); sc = i64(0)
rc = L.kgma_host_semiglobal_cigar(a, i64(len(a)), b, i64(len(b)), C.c_int32(-69), C.c_int32(-1), cig, i64(4096), C.byref(sc))
assert rc == 0 and len(cig.value) > 0
rc = L.kgma_host_semiglobal_cigar(a, i64(len(a)), b, i64(len(b)), C.c_int32(-69), C.c_int32(-1), cig, i64(3), C.byref(sc))
print("aligner ok (small buffer rc=%d)" % rc)
# 3. chain walk with HOSTILE chunk records: every chunk detailed with raw steps although only a few windows are wanted, and an
#    interval that no raw step covers -- the walk must stop, not write past `out`
steps = L.kgma_chain_chunk_steps()
nk = 284
n_valid = 5000
n_pos = n_valid + nk - 1
nb = (n_pos + 63) // 64
n_chunks = (nb + steps - 1) // steps
chunk_dt = np.dtype([("A0", np.int64), ("info", np.uint32), ("raw", np.uint32)])
chunks = np.zeros(n_chunks, dtype=chunk_dt)
pool = np.zeros(n_chunks * steps * 33 + 64, dtype=chunk_dt)
cur = 0
for c in range(n_chunks):
    st = min(steps, nb - c * steps)
    chunks[c]["A0"] = 0; chunks[c]["info"] = 1 | (0 << 2) | (1 << 11); chunks[c]["raw"] = cur     # leading run of 0 steps, detailed
    ent = cur; cur += st
    for s_ in range(st):
        pool[ent + s_]["info"] = 1 | (1 << 2) | (1 << 10); pool[ent + s_]["raw"] = cur; cur += 32
win0 = np.array([1], dtype=np.int64); nv = np.array([n_valid], dtype=np.int32); cb = np.array([0], dtype=np.int64); D0 = np.array([1000], dtype=np.int64)
for lo_, hi_ in (([10], [12]), ([10, 4000], [12, 4100]), ([1], [1])):
    lo = np.array(lo_, dtype=np.int64); hi = np.array(hi_, dtype=np.int64)
    tot = int((hi - lo + 1).sum())
    out = np.zeros(tot); n = i64(0); md = dbl(0)
    rc = L.kgma_host_chain_walk(dbl(1000 / 1008.0), dbl(1008.0), C.c_int32(nk), i64(1), P(win0, i64), P(nv, C.c_int32), P(cb, i64), P(D0, i64),
                                chunks.ctypes.data_as(C.c_void_p), i64(n_chunks), pool.ctypes.data_as(C.c_void_p), i64(cur), P(lo, i64), P(hi, i64), i64(len(lo_)),
                                P(out, dbl), i64(tot), C.byref(n), C.byref(md))
    print("walk", lo_, hi_, "rc", rc, "n", n.value)
# a remote hot list that disagrees: wanted windows in steps that are NOT raw -> regular chunks only
chunks2 = np.zeros(n_chunks, dtype=chunk_dt)
for c in range(n_chunks):
    st = min(steps, nb - c * steps)
    chunks2[c]["info"] = 1 | (st << 2)
lo = np.array([100], dtype=np.int64); hi = np.array([4000], dtype=np.int64)
out = np.zeros(3901); n = i64(0); md = dbl(0)
rc = L.kgma_host_chain_walk(dbl(1000 / 1008.0), dbl(1008.0), C.c_int32(nk), i64(1), P(win0, i64), P(nv, C.c_int32), P(cb, i64), P(D0, i64),
                            chunks2.ctypes.data_as(C.c_void_p), i64(n_chunks), pool.ctypes.data_as(C.c_void_p), i64(cur), P(lo, i64), P(hi, i64), i64(1),
                            P(out, dbl), i64(3901), C.byref(n), C.byref(md))
print("walk without the raw steps: rc", rc, "(must not be 0)")
assert rc != 0
print("ASAN RUN DONE")
