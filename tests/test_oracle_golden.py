"""Pins the CPU oracle (oracle/kgma_oracle.c) against the reference's own golden vectors."""
import os

import numpy as np
import pytest

from kmergma_amd import headers, refprep
from oracle import oracle as orc


def _hdr(recs, h):
    return headers.single_header(recs[h["contig"]].identifier, h["dist"], h["lo"], h["hi"], h["genome_pos"])


def test_kmer_known_answers(golden):
    g = golden["kmers"]
    seq = g["test_seq"].encode()
    assert orc.kmer_count(seq, 1).tolist() == g["kmer_count_k1"]
    assert orc.kmer_count(seq, 2).tolist() == g["kmer_count_k2"]
    for case in g["kmer_dist"]:
        a = seq * case["repeat"] + case["mid1"].encode() + seq * case["repeat"]
        b = seq * case["repeat"] + case["mid2"].encode() + seq * case["repeat"]
        assert orc.kmer_dist_seq(a, b, case["k"]) == case["expected"]
    # host-side helpers restating the same functions
    assert refprep.kmer_count(seq, 2).tolist() == g["kmer_count_k2"]
    assert refprep.as_UInt(seq) == g["as_UInt"]
    assert refprep.as_kmer(g["as_kmer"]["value"], g["as_kmer"]["len"]).decode() == g["as_kmer"]["expected"]


def test_n_maps_to_t_and_case_insensitive():
    assert orc.kmer_count(b"acgtn", 1).tolist() == [1, 1, 1, 2]
    with pytest.raises(orc.OracleError) as e:
        orc.kmer_count(b"ACGRT", 1)
    assert e.value.position == 4


def test_single_no_align_G1(golden, alp_ref, loci):
    g = golden["scan"]["single_no_align"]
    hits, _ = orc.single_scan([r.sequence for r in loci], alp_ref["RV"], 6, alp_ref["ws"], g["thr"], g["buff"])
    assert len(hits) == g["n_hits"]
    for idx, expected in g["headers"].items():
        assert _hdr(loci, hits[int(idx) - 1]) == expected


def test_single_dists_G3(golden, alp_ref, loci):
    g = golden["scan"]["single_dists"]
    hits, d = orc.single_scan([r.sequence for r in loci], alp_ref["RV"], 6, alp_ref["ws"], g["thr"], g["buff"],
                              return_dists=True)
    assert len(d) == g["n_dists"]
    assert round(float(d.mean())) == g["round_mean"]
    assert len(hits) == g["n_hits"]
    assert _hdr(loci, hits[0]) == g["headers"]["1"]
    assert _hdr(loci, hits[-1]) == g["headers"]["3"]


def test_append_hit_header(golden):
    g = golden["scan"]["append_hit"]
    assert headers.single_header(g["id"], g["dist"], g["range"][0], g["range"][1], g["genome_pos"]) == g["header"]
    assert g["seq"][g["range"][0] - 1:g["range"][1]] == g["body"]


def test_omn_dist_and_kfv_G5(golden, alp_clusters, alp_locus):
    """Dist / KFV fields of the cluster-mode expectations do not depend on BioAlignments."""
    g = golden["scan"]["omn_buff200"]
    hits, _ = orc.omn_scan([r.sequence for r in alp_locus], alp_clusters["KFVs"], 6, alp_clusters["ws"],
                           g["thr_vec"], g["buff"])
    assert len(hits) == g["n_hits"]
    got = [[headers.julia_round2(h["dist"]), h["kfv"]] for h in hits]
    assert got == g["dist_kfv"]
    assert all(h["genome_pos"] == 0 for h in hits)
    # pre-alignment CMIs derived in SURVEY.md Appendix B (G6)
    assert [h["cmi"] for h in hits] == [6851, 23690, 33843]


def test_exact_integer_restatement_matches_float(alp_ref, loci):
    """Appendix A.4: D/(2kN^2) equals the Float64 chain within 1e-11, hits identical."""
    seqs = [r.sequence for r in loci]
    k, W, N = 6, alp_ref["ws"], alp_ref["N"]
    hits, d = orc.single_scan(seqs, alp_ref["RV"], k, W, 30, 50, return_dists=True)
    T = orc.int_threshold(30, k, N)
    assert T == 30 * 2 * k * N * N
    hi, D, D1 = orc.single_scan_int(seqs, alp_ref["S"], N, k, W, T, 50, return_D=True)
    assert [(h["contig"], h["cmi"], h["lo"], h["hi"], h["genome_pos"]) for h in hits] == \
           [(h["contig"], h["cmi"], h["lo"], h["hi"], h["genome_pos"]) for h in hi]
    dd = D / (2.0 * k * N * N)
    assert np.abs(dd - d).max() < 1e-10
    for a, b in zip(hits, hi):
        assert abs(a["dist"] - b["D"] / (2.0 * k * N * N)) < 1e-6 * a["dist"]


def test_int_threshold_guard_band():
    # T = ceil(thr * 2kN^2 * (1 - 2^-30)): exact away from the distance lattice, and a window whose
    # distance is within rounding noise of thr counts as "not below"
    k, N = 6, 84
    s = 2 * k * N * N
    assert orc.int_threshold(30.0, k, N) == 30 * s                       # d == thr is not below
    assert orc.int_threshold(np.nextafter(30.0, 31), k, N) == 30 * s     # 1 ulp above a lattice value: noise
    assert orc.int_threshold(np.nextafter(30.0, 0), k, N) == 30 * s
    assert orc.int_threshold(30.0 * (1 + 2.0 ** -29), k, N) == 30 * s + 1
    assert orc.int_threshold(0.0, k, N) == 0
    assert orc.int_threshold(33.5, k, N) == int(33.5 * s)
    assert orc.int_threshold(6.4, 5, 1) == 64                            # 6.4 > 64/10 exactly, but only by rounding
    assert orc.int_threshold(33.3, k, N) == int(np.ceil(33.3 * s))


def test_short_contig_genome_pos_quirk(alp_ref):
    """Single engine: short records do not advance genome_pos (GenomeMiner.jl:37-39)."""
    rng = np.random.default_rng(1)
    short = bytes(rng.choice(list(b"ACGT"), size=100).tolist())
    long_ = bytes(rng.choice(list(b"ACGT"), size=5000).tolist())
    hits, _ = orc.single_scan([short, long_, long_], alp_ref["RV"], 6, alp_ref["ws"], 45, 50)
    gp = sorted({h["genome_pos"] for h in hits})
    assert set(gp) <= {0, 5000}
