"""CPU-side checks of the product library: it loads, exports every symbol include/kgma.h declares,
and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from kmergma_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "kgma.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kgma_[a-z_0-9]+)\s*\(", text)) - {"kgma_align_fn"})


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libkgma.so does not export {s}"
    assert set(_lib.EXPORTS) == set(syms)


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.KgmaHit) == 64
    assert ctypes.sizeof(_lib.KgmaDip) == 64
    assert ctypes.sizeof(_lib.KgmaStats) == 160


def test_version_and_status_strings():
    L = _lib.load()
    assert L.kgma_version() >= 1
    assert L.kgma_status_string(_lib.KGMA_E_BADBASE).decode().startswith("residue outside")


def test_no_cpu_fallback_without_gpu():
    from tests.conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.KgmaError) as e:
        _lib.Context(0)
    assert e.value.status == _lib.KGMA_E_NODEVICE
    from kmergma_amd import api
    with pytest.raises(_lib.KgmaError):
        api.ac_gma_testing(genome_path=os.path.join(ROOT, "tests", "data", "Alp_V_locus.fasta"),
                           refVec=[0.0] * 4096, do_align=False, resultVec=[])
