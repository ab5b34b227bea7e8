"""Host-side reference preparation vs the reference's known answers (test-KmerGMA.jl:48-111)."""
import os

import numpy as np

from kmergma_amd import refprep


def test_gen_ref_ws_cons(golden, data_dir):
    g = golden["refprep"]
    tf = os.path.join(data_dir, g["ref_fasta"])
    kfv, ws, cons, maxlen = refprep.gen_ref_ws_cons(tf, 1, get_maxlen=True)
    assert kfv.tolist() == g["gen_ref_ws_cons_k1"]["kfv"]
    assert ws == g["gen_ref_ws_cons_k1"]["windowsize"]
    assert maxlen == g["gen_ref_ws_cons_k1"]["maxlen"]
    assert cons.decode() == g["test_consensus_seq"]
    assert refprep.gen_ref_ws_cons(tf, 2)[0].tolist() == g["gen_ref_ws_cons_k2_kfv"]
    assert refprep.gen_ref_ws_cons(tf, 6)[0][4:10].tolist() == g["gen_ref_ws_cons_k6_kfv_5_10"]
    # integer form: KFV == S * (1/N)
    kfv6, _, _, (S, N) = refprep.gen_ref_ws_cons(tf, 6, return_int=True)
    assert N == 84 and np.array_equal(kfv6, S * (1.0 / N))


def test_get_cluster_index(golden):
    for inp, cut, exp in golden["refprep"]["get_cluster_index"]:
        assert refprep.get_cluster_index(inp, cut) == exp


def test_cluster_ref_api(golden, data_dir):
    g = golden["refprep"]
    c = g["cluster_ref_API_k1"]
    tf = os.path.join(data_dir, g["ref_fasta"])
    a = refprep.cluster_ref_API(tf, 1, cutoffs=c["cutoffs"], include_avg=False)
    assert [x.tolist() for x in a[0]] == c["kfvs_no_avg"]
    assert a[1] == c["windowsizes_no_avg"]
    assert len(a[2]) == 5 and a[2][0][:4].decode() == c["first_consensus_prefix"]
    assert a[3] == [False] * 5
    b = refprep.cluster_ref_API(tf, 1, cutoffs=c["cutoffs"])
    assert [x.tolist() for x in b[0]] == c["kfvs_no_avg"] + [c["avg_kfv"]]
    assert b[1] == c["windowsizes_with_avg"]
    assert b[3] == [False] * 6


def test_eliminate_null_params(golden, data_dir):
    g = golden["refprep"]
    kf = [np.array([1.0]), np.array([1.3])]
    out = refprep.eliminate_null_params(kf, [8, 9], [b"ATGCATGC", b"ATGCATGCY"], [False, True])
    assert out[1] == [8] and out[2] == [b"ATGCATGC"] and len(out[0]) == 1
    c = g["cluster_ref_API_k6"]
    tf = os.path.join(data_dir, g["ref_fasta"])
    RVs, ws, cons, inv = refprep.cluster_ref_API(tf, 6, cutoffs=c["cutoffs"])
    RVs, ws, cons = refprep.eliminate_null_params(RVs, ws, cons, inv)
    assert ws == c["windowsizes_after_eliminate"]
    assert len(RVs) == len(cons) == c["n_kfvs"]


def test_profile_consensus():
    p = refprep.Profile(8)
    p.add(b"ATGCATGC")
    assert [v.tolist() for v in p.vecs] == [[1, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0, 1],
                                             [0, 0, 1, 0, 0, 0, 1, 0], [0, 1, 0, 0, 0, 1, 0, 0]]
    p.lengthen(9)
    assert p.len == 9
    p.add(b"ATGCATGG"); p.add(b"ATGCATGG")
    assert p.consensus()[:8] == b"ATGCATGG"
