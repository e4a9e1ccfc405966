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


def test_every_entry_point_is_a_function_try_block():
    """"never aborts" (include/reflexiv_hip.h): every extern "C" definition with a body that can throw ends in RFX_API_CATCH --
    a C++ exception (std::bad_alloc from a vector, std::system_error from a thread) becomes RFX_E_HOST with its what() in
    rfx_last_error(), not std::terminate inside the JVM that loaded the JNI shim."""
    src_dir = os.path.join(ROOT, "reflexiv_amd", "csrc")
    exported = set(declared_functions())
    seen = set()
    for f in sorted(os.listdir(src_dir)):
        if not f.endswith(".hip"):
            continue
        lines = open(os.path.join(src_dir, f)).read().split("\n")
        for i, ln in enumerate(lines):
            m = re.match(r"^(?:int|void|int64_t|void \*|const char \*) ?\*?(rfx_[a-z0-9_]+)\(", ln)
            if not m or m.group(1) not in exported:
                continue
            name = m.group(1)
            seen.add(name)
            j = i
            while not lines[j].rstrip().endswith(("{", "}")):
                j += 1
            if lines[j].rstrip().endswith("}") and j == i:
                body = ln[ln.index("{"):]
                assert not re.search(r"\bnew\b|std::|\.push_back|\.alloc\(", body), (f, name, "one-line body that may throw")
                continue
            if name == "rfx_last_error":                    # (catches inside)
                continue
            assert lines[j].rstrip().endswith("try {"), (f, name)
            e = j + 1
            while not lines[e].startswith("}"):
                e += 1
            assert "RFX_API_CATCH" in lines[e], (f, name)
    assert seen == exported, sorted(exported - seen)


def test_exception_barrier_turns_a_cpp_exception_into_a_status(tmp_path, so):
    """The barrier itself, exercised on the CPU: a probe entry point written like the library's (function-try-block +
    RFX_API_CATCH) throws std::bad_alloc, std::length_error and a non-std object; each comes back as RFX_E_HOST with the
    text in the context's last_error."""
    import subprocess
    probe = tmp_path / "probe.cpp"
    probe.write_text(r'''
#include <new>
#include <stdexcept>
#include <vector>
#include "rfx_internal.h"
extern "C" int probe_throw(rfx_ctx *ctx, int what) try {
    if (what == 0) throw std::bad_alloc();
    if (what == 1) { std::vector<int> v; v.reserve((size_t)-1); }
    if (what == 2) throw 42;
    return RFX_OK;
} RFX_API_CATCH(ctx)
extern "C" rfx_ctx *probe_ctx() { return new rfx_ctx(); }
extern "C" const char *probe_err(rfx_ctx *c) { return c->last_error.c_str(); }
''')
    out = tmp_path / "libprobe.so"
    src_dir = os.path.join(ROOT, "reflexiv_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + src_dir,
                    str(probe), "-o", str(out), "-L" + os.path.join(ROOT, "reflexiv_amd"), "-lreflexiv_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "reflexiv_amd")], check=True, capture_output=True)
    P = ctypes.CDLL(str(out))
    P.probe_ctx.restype = ctypes.c_void_p
    P.probe_err.restype = ctypes.c_char_p
    P.probe_err.argtypes = [ctypes.c_void_p]
    P.probe_throw.argtypes = [ctypes.c_void_p, ctypes.c_int]
    c = P.probe_ctx()
    assert P.probe_throw(c, 3) == _lib.RFX_OK
    for what, text in ((0, b"bad_alloc"), (1, b"vector"), (2, b"unknown C++ exception")):
        assert P.probe_throw(c, what) == _lib.RFX_E_HOST == -7
        assert text in P.probe_err(c) and b"probe_throw" in P.probe_err(c), P.probe_err(c)
    assert P.probe_throw(None, 0) == _lib.RFX_E_HOST           # no context to carry the text: still a status
