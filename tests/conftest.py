import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kmergma.jl_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

DATA = os.path.join(ROOT, "tests", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


@pytest.fixture(scope="session")
def data_dir():
    return DATA


@pytest.fixture(scope="session")
def golden():
    out = {}
    for name in ("kmers", "refprep", "scan"):
        with open(os.path.join(GOLDEN, name + ".json")) as fh:
            out[name] = json.load(fh)
    return out


@pytest.fixture(scope="session")
def alp_ref(data_dir):
    """k=6 single-KFV inputs from the alpaca IGHV fixture: dict(RV, ws, cons, S, N)."""
    from kmergma_amd import refprep
    RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(os.path.join(data_dir, "Alp_V_ref.fasta"), 6, return_int=True)
    return dict(RV=RV, ws=ws, cons=cons, S=S, N=N, k=6)


@pytest.fixture(scope="session")
def alp_clusters(data_dir):
    """k=6 cluster inputs (include_avg=False): 5 KFVs."""
    from kmergma_amd import refprep
    KFVs, ws, cons, inv, ints = refprep.cluster_ref_API(os.path.join(data_dir, "Alp_V_ref.fasta"), 6,
                                                        cutoffs=[7, 12, 20, 25], include_avg=False, return_int=True)
    KFVs, ws, cons, ints = refprep.eliminate_null_params(KFVs, ws, cons, inv, ints)
    return dict(KFVs=KFVs, ws=ws, cons=cons, S=[s for s, _ in ints], N=[n for _, n in ints], k=6)


@pytest.fixture(scope="session")
def loci(data_dir):
    from kmergma_amd import fasta
    return fasta.read_fasta(os.path.join(data_dir, "Loci.fasta"))


@pytest.fixture(scope="session")
def alp_locus(data_dir):
    from kmergma_amd import fasta
    return fasta.read_fasta(os.path.join(data_dir, "Alp_V_locus.fasta"))
