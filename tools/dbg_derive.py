import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "kmergma.jl_amd"))
from kmergma_amd import _lib, refprep, fasta
from oracle import oracle as orc
from tests.helpers import make_genome
data_dir = os.path.join(R, "tests", "data")
KFVs, ws, cons, inv, ints = refprep.cluster_ref_API(os.path.join(data_dir, "Alp_V_ref.fasta"), 6, cutoffs=[7, 12, 20, 25], include_avg=False, return_int=True)
KFVs, ws, cons, ints = refprep.eliminate_null_params(KFVs, ws, cons, inv, ints)
S = [s for s, _ in ints]; N = [n for _, n in ints]
k = 6
print("ws", ws, "N", N)
genes = [r.sequence.upper() for r in fasta.read_fasta(os.path.join(data_dir, "Alp_V_ref.fasta"))]
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(seed)
maxws = max(ws); P = 32768
lengths = [maxws + k - 2, maxws + k - 1, maxws + k, 200, P + maxws + k - 2, P + maxws + k, 90011, 2 * P + 5000, 6]
contigs, _ = make_genome(rng, lengths, genes, n_plants_per_mb=120)
thr = [37, 33, 38, 34, 28]
ctx = _lib.Context(0)
ctx.set_refs(k, KFVs, ws, thr, N)
gen = ctx.genome_from_host(contigs)
ctx.scan(gen, _lib.MODE_OMN, 50, 1234, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
dists = [ctx.dists(j + 1) for j in range(len(ws))]
T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
ohi, oD = orc.omn_scan_int(contigs, S, N, k, ws, T, 50, 1234, return_D=True)
offs = np.cumsum([0] + [max(0, len(c) - maxws + 1) for c in contigs])
print("record window offsets", offs)
for j in range(len(ws)):
    exp = oD[j] / (2.0 * k * N[j] ** 2)
    bad = np.nonzero(dists[j] != exp)[0]
    print("kfv", j, "ws", ws[j], "n bad", len(bad), "first", bad[:10], "last", bad[-5:])
    for b in bad[:5]:
        print("   ", b, dists[j][b] * (2.0 * k * N[j] ** 2), oD[j][b])
