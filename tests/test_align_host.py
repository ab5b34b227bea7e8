"""Host-side alignment helper vs the reference's expectations that go through BioAlignments
(test/test_folder/test-KmerGMA.jl:128-145 and the loci of :179-193).  Runs on the CPU."""
import os

from kmergma_amd import align, fasta, refprep


def test_cigar_to_unitrange_goldens(golden):
    for case in golden["scan"]["alignment"]["cigar_to_UnitRange"]:
        got = align.align_range(case["a"].encode(), case["b"].encode(), case["gap_open"], case["gap_extend"])
        assert list(got) == case["expected"]
    assert align.cigar_to_UnitRange("5D8=5D") == (6, 13)
    assert align.cigar_to_UnitRange("5D4=2D4=5D") == (6, 15)
    # quirks of src/Alignment.jl:13-30: last op dropped, first op taken whatever its type
    assert align.cigar_to_UnitRange("8=") == (1, 0)
    assert align.cigar_to_UnitRange("3=2I3=4D") == (4, 8)


def test_align_unitrange_golden(golden, data_dir):
    g = golden["scan"]["alignment"]["align_unitrange"]
    cons = golden["refprep"]["test_consensus_seq"].encode()
    rec = fasta.read_fasta(os.path.join(data_dir, g["genome"]))[g["record"] - 1]
    lo, hi = g["range"]
    a, b = align.align_range(cons[:g["windowsize"]], rec.sequence[lo - 1:hi], -69, -1)
    assert [max(1, lo + a - 1), min(lo + b - 1, g["seq_len"])] == g["expected"]


def test_fixture_hit_loci_through_aligner(golden, data_dir, loci):
    """Pre-alignment candidates (golden G4) -> aligner -> the loci of test-KmerGMA.jl:189."""
    g = golden["scan"]["single_align"]
    RV, W, cons = refprep.gen_ref_ws_cons(os.path.join(data_dir, "Alp_V_ref.fasta"), 6)
    pre = [(0, 8498, 8886, 0), (0, 20380, 20768, 0), (2, 640, 1028, 221227), (2, 12746, 13134, 221227),
           (3, 6807, 7195, 444023), (3, 23864, 24252, 444023), (3, 33800, 34188, 444023)]
    out = []
    for c, lo, hi, gp in pre:
        a, b = align.align_range(cons[:W], loci[c].sequence[lo - 1:hi], g["gap_open"], g["gap_extend"])
        out.append(max(1, lo + a - 1) + gp)
    assert out == g["hit_loci"]
