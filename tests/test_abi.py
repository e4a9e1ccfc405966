"""CPU-side checks of the drop-in boundary: the shared library loads and exports every
function include/reflexiv_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

import reflexiv_amd
from reflexiv_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "reflexiv_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rfx_[a-z0-9_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def so():
    return ctypes.CDLL(_lib.build())


def test_header_and_binding_list_agree():
    assert declared_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(so):
    for name in declared_functions():
        assert hasattr(so, name), name


def test_version_and_defaults(so):
    assert so.rfx_version() >= 100
    p = reflexiv_amd.default_params()
    # U/DefaultParam.java:74,103-107,114-115
    assert (p.k, p.min_cov, p.max_cov, p.min_error_cov, p.min_contig, p.min_iter, p.max_iter) == \
        (31, 2, 10_000_000, 8, 500, 15, 150)


def test_kmers_per_read_rule(so):
    """skip rule len - k - endClip <= 1 (P/ReflexivMain.java:3020) as host arithmetic."""
    L = _lib.lib()
    assert [L.rfx_kmers_per_read(n, 31, 0, 0) for n in (30, 31, 32, 33, 100, 150)] == [0, 0, 0, 3, 70, 120]
    assert L.rfx_kmers_per_read(150, 31, 5, 10) == 105
    assert L.rfx_kmers_per_read(40, 31, 41, 0) == 0


def test_no_silent_cpu_fallback():
    """Without a gfx950 GPU a context cannot be created -- it must raise, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(reflexiv_amd.RfxError):
        reflexiv_amd.Reflexiv()


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under reflexiv_amd/ may import, include or link it."""
    pkg = os.path.join(ROOT, "reflexiv_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle|#include\s*[<\"].*oracle|liborc|orc_[a-z]+\s*\(", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), f


def test_rccl_stand_in_of_the_multirank_tests_exports_what_the_library_binds():
    """tests/fake_rccl (test infrastructure: several ranks on one GPU) must serve every RCCL entry point rfx_comm.hip
    looks up with dlsym -- read from the source, so a new binding cannot be forgotten there."""
    import ctypes
    import re
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.run(["make", "-s", "-C", os.path.join(here, "fake_rccl")], check=True, capture_output=True)
    shim = ctypes.CDLL(os.path.join(here, "fake_rccl", "libfake_rccl.so"))
    src = open(os.path.join(here, "..", "reflexiv_amd", "csrc", "rfx_comm.hip")).read()
    names = re.findall(r'sym\("(nccl[A-Za-z]+)"\)', src)
    assert len(names) >= 10
    for n in names:
        assert hasattr(shim, n), n
