"""ctypes/numpy front-end of the CPU oracle (oracle/reflexiv_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by reflexiv_amd/.  See reflexiv_oracle.h
for what pins it (docs/example.html known answer) and what does not.
"""
from __future__ import annotations

import ctypes as C
import gzip
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TWIN_DS, TWIN_RDD = 0, 1


class Params(C.Structure):
    """orc_params (mirrors U/DefaultParam.java:74-120 for the hot path)."""
    _fields_ = [(n, C.c_int32) for n in (
        "k", "min_cov", "max_cov", "min_error_cov", "min_contig", "min_iter", "max_iter",
        "front_clip", "end_clip", "partitions", "twin", "coalesce", "extras")]


class _Records(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", C.c_void_p), ("marker", C.c_void_p),
                ("ext_off", C.c_void_p), ("ext", C.c_void_p), ("left", C.c_void_p),
                ("right", C.c_void_p)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("reflexiv_oracle.c", "reflexiv_dedup.c", "reflexiv_dynamic.c", "reflexiv_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_fastq_group.restype = C.c_int64
        L.orc_fastq_only_seq.restype = C.c_int64
        L.orc_extract_canon.restype = C.c_int64
        L.orc_count_filter.restype = C.c_int64
        L.orc_extract_canon_w.restype = C.c_int64
        L.orc_count_filter_w.restype = C.c_int64
        L.orc_revcomp.restype = C.c_uint64
        L.orc_revcomp.argtypes = [C.c_uint64, C.c_int]
        L.orc_fork_filter_forward.restype = C.c_int64
        L.orc_fork_filter_reflected.restype = C.c_int64
        L.orc_extend_pass.restype = C.c_int64
        L.orc_contigs_text.restype = C.c_int64
        L.orc_assemble_from_counts.restype = C.c_int64
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_count_reads_omp.restype = C.c_int64
        L.orc_count_reads_range_omp.restype = C.c_int64
        L.orc_count_reads_w2_range_omp.restype = C.c_int64
        L.orc_dyn_extend_pass.restype = C.c_int64
        L.orc_dyn_random_reflection.restype = C.c_int64
        L.orc_dedup_contigs.restype = C.c_int64
        L.orc_dedup_text.restype = C.c_int64
        for f in ("orc_double_w", "orc_key_filter_w", "orc_flip_all_w"):
            getattr(L, f).restype = C.c_int64
        for f in ("orc_fork_filter_forward_w", "orc_fork_filter_reflected_w", "orc_extend_pass_w",
                  "orc_contigs_text_w", "orc_assemble_from_counts_w"):
            getattr(L, f).restype = C.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params(**kw) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


# ------------------------------------------------------------------ records

@dataclass
class Records:
    """Flat SoA in the reference's record layout."""
    key: np.ndarray       # uint64 [n] (k <= 32) or [n, kw] (k > 32: kw = (k-2)//31+1 words of 31 bases)
    marker: np.ndarray    # int32  [n]
    ext_off: np.ndarray   # int64  [n+1]
    ext: np.ndarray       # uint64 [ext_off[n]]
    left: np.ndarray      # int32  [n]
    right: np.ndarray     # int32  [n]

    @property
    def n(self) -> int:
        return int(self.key.shape[0])

    @staticmethod
    def from_single(key, marker, ext, left, right) -> "Records":
        n = len(key)
        key = np.ascontiguousarray(key, np.uint64)
        if key.ndim == 2 and key.shape[1] == 1:
            key = key.reshape(-1)
        return Records(key, np.ascontiguousarray(marker, np.int32),
                       np.arange(n + 1, dtype=np.int64), np.ascontiguousarray(ext, np.uint64),
                       np.ascontiguousarray(left, np.int32), np.ascontiguousarray(right, np.int32))

    def tuple_list(self):
        kk = self.key if self.key.ndim == 1 else [tuple(int(x) for x in r) for r in self.key]
        return [(kk[i] if self.key.ndim > 1 else int(kk[i]), int(self.marker[i]),
                 tuple(int(x) for x in self.ext[self.ext_off[i]:self.ext_off[i + 1]]),
                 int(self.left[i]), int(self.right[i])) for i in range(self.n)]


# ---------------------------------------------------------------- operators

def fastq_group(text: bytes):
    """a-1 FastqFilterWithQual (P/ReflexivMain.java:3089-3113) -> (seq_off, seq_len)."""
    buf = np.frombuffer(text, dtype=np.uint8)
    n = lib().orc_fastq_group(_p(buf), C.c_int64(len(text)), None, None, C.c_int64(0))
    off = np.empty(n, np.int64)
    ln = np.empty(n, np.int32)
    lib().orc_fastq_group(_p(buf), C.c_int64(len(text)), _p(off), _p(ln), C.c_int64(n))
    return off, ln


def fastq_only_seq(text: bytes):
    """DSFastqFilterOnlySeq (P/ReflexivDataFrameCounter.java:238-290) -> (seq_off, seq_len)."""
    buf = np.frombuffer(text, dtype=np.uint8)
    n = lib().orc_fastq_only_seq(_p(buf), C.c_int64(len(text)), None, None, C.c_int64(0))
    off = np.empty(n, np.int64)
    ln = np.empty(n, np.int32)
    lib().orc_fastq_only_seq(_p(buf), C.c_int64(len(text)), _p(off), _p(ln), C.c_int64(n))
    return off, ln


def load_fastq(paths):
    """Concatenate the reads of FASTQ(.gz) files -> (bases uint8[], read_off int64[n+1])."""
    seqs, lens = [], []
    for path in paths:
        opener = gzip.open if path.endswith(".gz") else open
        with opener(path, "rb") as fh:
            text = fh.read()
        off, ln = fastq_group(text)
        tb = np.frombuffer(text, dtype=np.uint8)
        for o, l in zip(off, ln):
            seqs.append(tb[o:o + l])
        lens.extend(int(x) for x in ln)
    read_off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(np.asarray(lens, np.int64), out=read_off[1:])
    bases = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases), read_off


def extract_canon(bases: np.ndarray, read_off: np.ndarray, k=31, front_clip=0, end_clip=0) -> np.ndarray:
    bases = np.ascontiguousarray(bases, np.uint8)
    read_off = np.ascontiguousarray(read_off, np.int64)
    nr = len(read_off) - 1
    n = lib().orc_extract_canon(_p(bases), _p(read_off), C.c_int64(nr), k, front_clip, end_clip,
                                None, C.c_int64(0))
    out = np.empty(n, np.uint64)
    lib().orc_extract_canon(_p(bases), _p(read_off), C.c_int64(nr), k, front_clip, end_clip,
                            _p(out), C.c_int64(n))
    return out


def count_filter(kmers: np.ndarray, min_cov=2, max_cov=10_000_000, twin=TWIN_DS):
    """-> (keys uint64[], counts int32[], n_distinct); kmers is not modified."""
    work = np.array(kmers, dtype=np.uint64, copy=True)
    n = len(work)
    keys = np.empty(n, np.uint64)
    counts = np.empty(n, np.int32)
    nd = C.c_int64(0)
    m = lib().orc_count_filter(_p(work), C.c_int64(n), min_cov, max_cov, twin,
                               _p(keys), _p(counts), C.c_int64(n), C.byref(nd))
    return keys[:m].copy(), counts[:m].copy(), int(nd.value)


def set_threads(t: int):
    """threads for the CPU-baseline leg (1 = strictly serial, the default and what the parity tests use)"""
    lib().orc_set_threads(int(t))


def host_cores() -> int:
    return int(lib().orc_host_cores())


def count_reads_omp(bases, read_off, k=31, min_cov=2, max_cov=10_000_000, twin=TWIN_DS, front_clip=0, end_clip=0,
                    cap=None, buckets=(0, 4096)):
    """extract + reduceByKey + filter over the threads of set_threads() -> (keys, counts, n_distinct, n_instances).
    buckets = (lo, hi): only the k-mers whose top 12 bits fall in [lo, hi) (passes over shares of the k-mer space).
    k = 33..63: two-word k-mers uint64[m, 2], int64 counts (ReflexivDataFrameCounter64's filters)."""
    bases = np.ascontiguousarray(bases, np.uint8)
    read_off = np.ascontiguousarray(read_off, np.int64)
    nr = len(read_off) - 1
    if cap is None:
        cap = max(1, int(read_off[-1] - read_off[0]))
    nd, ni = C.c_int64(0), C.c_int64(0)
    if k > 31:
        keys = np.empty((cap, 2), np.uint64); counts = np.empty(cap, np.int64)
        m = lib().orc_count_reads_w2_range_omp(_p(bases), _p(read_off), C.c_int64(nr), k, front_clip, end_clip, min_cov,
                                               max_cov, buckets[0], buckets[1], _p(keys), _p(counts), C.c_int64(cap),
                                               C.byref(nd), C.byref(ni))
        if m < 0:
            raise ValueError("k must be 33..63")
    else:
        keys = np.empty(cap, np.uint64); counts = np.empty(cap, np.int32)
        m = lib().orc_count_reads_range_omp(_p(bases), _p(read_off), C.c_int64(nr), k, front_clip, end_clip, min_cov, max_cov,
                                            twin, buckets[0], buckets[1], _p(keys), _p(counts), C.c_int64(cap),
                                            C.byref(nd), C.byref(ni))
    if m > cap:
        raise ValueError(f"cap {cap} < {m} survivors")
    return keys[:m].copy(), counts[:m].copy(), int(nd.value), int(ni.value)


def words_w(k: int) -> int:
    return k // 32 + 1


def extract_canon_w(bases: np.ndarray, read_off: np.ndarray, k=63, front_clip=0, end_clip=0) -> np.ndarray:
    """k > 31 (ReflexivDataFrameCounter64): -> uint64[n, W], W = k//32+1, last word right-aligned."""
    bases = np.ascontiguousarray(bases, np.uint8)
    read_off = np.ascontiguousarray(read_off, np.int64)
    nr = len(read_off) - 1
    W = words_w(k)
    n = lib().orc_extract_canon_w(_p(bases), _p(read_off), C.c_int64(nr), k, front_clip, end_clip, None, C.c_int64(0))
    if n < 0:
        raise ValueError("k must be > 32 and not a multiple of 32")
    out = np.empty((n, W), np.uint64)
    lib().orc_extract_canon_w(_p(bases), _p(read_off), C.c_int64(nr), k, front_clip, end_clip, _p(out), C.c_int64(n))
    return out


def count_filter_w(kmers: np.ndarray, k=63, min_cov=2, max_cov=10_000_000):
    """-> (keys uint64[m, W] ascending, counts int64[m], n_distinct); kmers is not modified."""
    W = words_w(k)
    work = np.array(kmers, dtype=np.uint64, copy=True).reshape(-1, W)
    n = len(work)
    keys = np.empty((n, W), np.uint64)
    counts = np.empty(n, np.int64)
    nd = C.c_int64(0)
    m = lib().orc_count_filter_w(_p(work), C.c_int64(n), k, min_cov, max_cov, _p(keys), _p(counts), C.c_int64(n),
                                 C.byref(nd))
    return keys[:m].copy(), counts[:m].copy(), int(nd.value)


def kmer_text_w(kmer: np.ndarray, k: int) -> str:
    kmer = np.ascontiguousarray(kmer, np.uint64)
    buf = C.create_string_buffer(k)
    lib().orc_kmer_text_w(_p(kmer), k, buf)
    return buf.raw.decode()


def revcomp(kmer: int, k: int) -> int:
    return int(lib().orc_revcomp(C.c_uint64(kmer), k))


def sub_words(k: int) -> int:
    """words of a (k-1)-mer key: 1 up to k = 32, (k-2)//31+1 beyond (subKmerBinarySlots)."""
    return int(lib().orc_sub_words(k))


def asm_words(k: int) -> int:
    """words of a k-mer in the assembler's 31-bases-per-word layout (kmerBinarySlotsAssemble)."""
    return int(lib().orc_asm_words(k))


def _kw_of(key: np.ndarray) -> int:
    return 1 if key.ndim == 1 else int(key.shape[1])


def _keybuf(n: int, kw: int) -> np.ndarray:
    return np.empty(n, np.uint64) if kw == 1 else np.empty((n, kw), np.uint64)


def kmer_binarize_w(kmer_text: str, count_text: str, k: int):
    """KmerBinarizer.call (P/ReflexivDSMain64.java:10772-10836) on one CSV row -> (words, cover)."""
    w = np.zeros(asm_words(k), np.uint64)
    c = C.c_int32(0)
    st = lib().orc_kmer_binarize_w(kmer_text.encode(), count_text.encode(), k, _p(w), C.byref(c))
    if st != 0:
        raise ValueError("k-mer text shorter than k")
    return w, int(c.value)


def counter_to_asm_w(kmers32: np.ndarray, k: int) -> np.ndarray:
    """counter layout uint64[n, k//32+1] -> assembler layout uint64[n, (k-1)//31+1] (the CSV round trip)."""
    kmers32 = np.ascontiguousarray(kmers32, np.uint64).reshape(-1, words_w(k))
    out = np.empty((len(kmers32), asm_words(k)), np.uint64)
    lib().orc_counter_to_asm_w(_p(kmers32), C.c_int64(len(kmers32)), k, _p(out))
    return out


def rc_expand_subkmer(keys, counts, k=31) -> Records:
    """k <= 31: keys uint64[n]; k > 31: keys uint64[n, (k-1)//31+1] in the assembler layout."""
    keys = np.ascontiguousarray(keys, np.uint64)
    counts = np.ascontiguousarray(counts, np.int32)
    n = len(counts)
    kw = sub_words(k)
    key = _keybuf(2 * n, kw); ext = np.empty(2 * n, np.uint64)
    marker = np.empty(2 * n, np.int32); left = np.empty(2 * n, np.int32); right = np.empty(2 * n, np.int32)
    lib().orc_rc_expand_subkmer_w(_p(keys), _p(counts), C.c_int64(n), k, _p(key), _p(marker), _p(ext),
                                  _p(left), _p(right))
    return Records.from_single(key, marker, ext, left, right)


def sort_perm(key: np.ndarray) -> np.ndarray:
    key = np.ascontiguousarray(key, np.uint64)
    perm = np.empty(len(key), np.int64)
    lib().orc_sort_perm_w(_p(key), C.c_int64(len(key)), _kw_of(key), _p(perm))
    return perm


def partition_starts(sorted_key: np.ndarray, P: int) -> np.ndarray:
    sorted_key = np.ascontiguousarray(sorted_key, np.uint64)
    st = np.empty(P + 1, np.int64)
    lib().orc_partition_starts_w(_p(sorted_key), C.c_int64(len(sorted_key)), _kw_of(sorted_key), P, _p(st))
    return st


def gather(r: Records, perm: np.ndarray) -> Records:
    perm = np.ascontiguousarray(perm, np.int64)
    n = len(perm)
    kw = _kw_of(r.key)
    out = Records(_keybuf(n, kw), np.empty(n, np.int32), np.empty(n + 1, np.int64),
                  np.empty(max(1, len(r.ext)), np.uint64), np.empty(n, np.int32), np.empty(n, np.int32))
    lib().orc_gather_w(_p(perm), C.c_int64(n), kw, _p(r.key), _p(r.marker), _p(r.ext_off), _p(r.ext),
                       _p(r.left), _p(r.right), _p(out.key), _p(out.marker), _p(out.ext_off),
                       _p(out.ext), _p(out.left), _p(out.right))
    out.ext = out.ext[:out.ext_off[n]].copy()
    return out


def sort_records(r: Records) -> Records:
    return gather(r, sort_perm(r.key))


def _fork(fn, r: Records, part_start, k, min_error_cov, twin):
    n = r.n
    P = len(part_start) - 1
    part_start = np.ascontiguousarray(part_start, np.int64)
    assert _kw_of(r.key) == sub_words(k), (r.key.shape, k)
    o = [_keybuf(n, _kw_of(r.key)), np.empty(n, np.int32), np.empty(n, np.uint64),
         np.empty(n, np.int32), np.empty(n, np.int32)]
    ops = np.empty(P + 1, np.int64)
    m = fn(_p(r.key), _p(r.marker), _p(r.ext), _p(r.left), _p(r.right), C.c_int64(n),
           _p(part_start), P, k, min_error_cov, twin,
           _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(o[4]), _p(ops))
    return Records.from_single(o[0][:m].copy(), o[1][:m].copy(), o[2][:m].copy(),
                               o[3][:m].copy(), o[4][:m].copy()), ops


def fork_filter_forward(r, part_start, k=31, min_error_cov=8, twin=TWIN_DS):
    return _fork(lib().orc_fork_filter_forward_w, r, part_start, k, min_error_cov, twin)


def fork_filter_reflected(r, part_start, k=31, min_error_cov=8, twin=TWIN_DS):
    return _fork(lib().orc_fork_filter_reflected_w, r, part_start, k, min_error_cov, twin)


def reflect_from_forward(r: Records, k=31) -> Records:
    n = r.n
    assert _kw_of(r.key) == sub_words(k), (r.key.shape, k)
    key = _keybuf(n, _kw_of(r.key)); marker = np.empty(n, np.int32); ext = np.empty(n, np.uint64)
    lib().orc_reflect_from_forward_w(_p(r.key), _p(r.ext), C.c_int64(n), k, _p(key), _p(marker), _p(ext))
    return Records.from_single(key, marker, ext, r.left.copy(), r.right.copy())


def random_reflection(r: Records, part_start, k=31) -> Records:
    part_start = np.ascontiguousarray(part_start, np.int64)
    out = Records.from_single(r.key.copy(), r.marker.copy(), r.ext.copy(), r.left.copy(), r.right.copy())
    assert _kw_of(r.key) == sub_words(k), (r.key.shape, k)
    lib().orc_random_reflection_w(_p(out.key), _p(out.marker), _p(out.ext), C.c_int64(out.n),
                                  _p(part_start), len(part_start) - 1, k)
    return out


def extend_pass(r: Records, part_start, k=31, twin=TWIN_DS, start_marker=2):
    """One extend pass over records already sorted by key -> (Records, out_part_start)."""
    n = r.n
    P = len(part_start) - 1
    part_start = np.ascontiguousarray(part_start, np.int64)
    words = max(1, int(r.ext_off[n]))
    assert _kw_of(r.key) == sub_words(k), (r.key.shape, k)
    o = Records(_keybuf(max(1, n), _kw_of(r.key)), np.empty(max(1, n), np.int32),
                np.empty(n + 1, np.int64), np.empty(words, np.uint64),
                np.empty(max(1, n), np.int32), np.empty(max(1, n), np.int32))
    ops = np.empty(P + 1, np.int64)
    m = lib().orc_extend_pass_w(_p(r.key), _p(r.marker), _p(r.ext_off), _p(r.ext), _p(r.left), _p(r.right),
                                C.c_int64(n), _p(part_start), P, k, twin, start_marker,
                                _p(o.key), _p(o.marker), _p(o.ext_off), _p(o.ext), _p(o.left), _p(o.right),
                                _p(ops))
    return Records(o.key[:m].copy(), o.marker[:m].copy(), o.ext_off[:m + 1].copy(),
                   o.ext[:o.ext_off[m]].copy(), o.left[:m].copy(), o.right[:m].copy()), ops


OP_EXTENDABLE_PAIRS, OP_UNEXTENDABLE, OP_FIRST_OF_KEY, OP_LONGER_OF_KEY = 1, 2, 3, 4


def _out_records(n_cap, w_cap, kw):
    return Records(_keybuf(max(1, n_cap), kw), np.empty(max(1, n_cap), np.int32), np.empty(n_cap + 1, np.int64),
                   np.empty(max(1, w_cap), np.uint64), np.empty(max(1, n_cap), np.int32), np.empty(max(1, n_cap), np.int32))


def _trim(o: Records, m: int) -> Records:
    return Records(o.key[:m].copy(), o.marker[:m].copy(), o.ext_off[:m + 1].copy(), o.ext[:o.ext_off[m]].copy(),
                   o.left[:m].copy(), o.right[:m].copy())


def double_records(r: Records, k: int) -> Records:
    """DSReflexivAndForwardKmer (P/ReflexivDSMain64.java:2126-3042): every record, then its other orientation"""
    kw = sub_words(k)
    o = _out_records(2 * r.n, 2 * int(r.ext_off[r.n]), kw)
    m = lib().orc_double_w(_p(r.key), _p(r.marker), _p(r.ext_off), _p(r.ext), _p(r.left), _p(r.right), C.c_int64(r.n), k,
                           _p(o.key), _p(o.marker), _p(o.ext_off), _p(o.ext), _p(o.left), _p(o.right))
    return _trim(o, m)


def key_filter(op: int, r: Records, part_start, k: int):
    """the four run filters of the k > 31 from-counts extras (OP_*) on records sorted by key -> (Records, out_part_start)"""
    kw = sub_words(k)
    part_start = np.ascontiguousarray(part_start, np.int64)
    P = len(part_start) - 1
    o = _out_records(r.n, int(r.ext_off[r.n]), kw)
    ops = np.empty(P + 1, np.int64)
    m = lib().orc_key_filter_w(op, _p(r.key), _p(r.marker), _p(r.ext_off), _p(r.ext), _p(r.left), _p(r.right), C.c_int64(r.n),
                               _p(part_start), P, k, _p(o.key), _p(o.marker), _p(o.ext_off), _p(o.ext), _p(o.left),
                               _p(o.right), _p(ops))
    return _trim(o, m), ops


def flip_all(r: Records, k: int, m: int) -> Records:
    """DSFilterUnExtendableKmerLeftEnds (m = 1) / ...RightEnds (m = 2): every record in orientation m"""
    kw = sub_words(k)
    o = _out_records(r.n, int(r.ext_off[r.n]), kw)
    n = lib().orc_flip_all_w(_p(r.key), _p(r.marker), _p(r.ext_off), _p(r.ext), _p(r.left), _p(r.right), C.c_int64(r.n), k, m,
                             _p(o.key), _p(o.marker), _p(o.ext_off), _p(o.ext), _p(o.left), _p(o.right))
    return _trim(o, n)


def contigs_text(r: Records, k=31, min_contig=500, twin=TWIN_DS):
    """k <= 31: the twin's header; k > 31 (P/ReflexivDSMain64.java:830-866): ">Contig-<len>-<idx>"."""
    n = r.n
    nc = C.c_int64(0)
    if k > 31:
        assert _kw_of(r.key) == sub_words(k), (r.key.shape, k)
        fn = lib().orc_contigs_text_w
        args = (_p(r.key), _p(r.marker), _p(r.ext_off), _p(r.ext), _p(r.left), _p(r.right),
                C.c_int64(n), k, min_contig)
    else:
        fn = lib().orc_contigs_text
        args = (_p(r.key), _p(r.marker), _p(r.ext_off), _p(r.ext), _p(r.left), _p(r.right),
                C.c_int64(n), k, min_contig, twin)
    ln = fn(*args, None, C.c_int64(0), C.byref(nc))
    buf = np.empty(max(1, ln), np.uint8)
    fn(*args, _p(buf), C.c_int64(ln), C.byref(nc))
    return bytes(buf[:ln]).decode(), int(nc.value)


def assemble_from_counts(keys, counts, prm: Params):
    """a-14 driver -> (text, n_contigs, trace[list of record counts per pass], final Records).
    k > 31: keys uint64[n, (k-1)//31+1] in the assembler layout, ReflexivDSMain64's driver."""
    keys = np.ascontiguousarray(keys, np.uint64)
    counts = np.ascontiguousarray(counts, np.int32)
    n = len(counts)
    wide = prm.k > 31
    kw = sub_words(prm.k)
    if wide:
        assert keys.shape == (n, asm_words(prm.k)), keys.shape
    trace = np.zeros(prm.max_iter + 8, np.int64)
    ntr = C.c_int64(0); nc = C.c_int64(0)
    rec = _Records()
    cap = 4 * (n + 16) * (prm.k + 8) + 1024
    buf = np.empty(cap, np.uint8)
    fn = lib().orc_assemble_from_counts_w if wide else lib().orc_assemble_from_counts
    ln = fn(_p(keys), _p(counts), C.c_int64(n), C.byref(prm), _p(buf), C.c_int64(cap), C.byref(nc),
            _p(trace), C.c_int64(len(trace)), C.byref(ntr), C.byref(rec))
    assert ln <= cap
    m = rec.n

    def arr(ptr, cnt, dt):
        if cnt == 0:
            return np.zeros(0, dt)
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(cnt * np.dtype(dt).itemsize,))
        return a.view(dt).copy()
    ext_off = arr(rec.ext_off, m + 1, np.int64)
    okey = arr(rec.key, m * kw, np.uint64)
    if kw > 1:
        okey = okey.reshape(m, kw)
    out = Records(okey, arr(rec.marker, m, np.int32), ext_off,
                  arr(rec.ext, int(ext_off[m]) if m else 0, np.uint64),
                  arr(rec.left, m, np.int32), arr(rec.right, m, np.int32))
    lib().orc_free_records(C.byref(rec))
    return bytes(buf[:ln]).decode(), int(nc.value), [int(x) for x in trace[:ntr.value]], out


# ------------------------------------------------------------ synthetic reads

def synth_genome(seed: int, genome_len: int) -> np.ndarray:
    g = np.empty((genome_len + 31) // 32, np.uint64)
    lib().orc_synth_genome(C.c_uint64(seed), C.c_int64(genome_len), _p(g))
    return g


def synth_reads(seed: int, genome: np.ndarray, genome_len: int, first_read: int, n_reads: int,
                read_len: int = 150, err_per_2_32: int = 21474836):
    """-> (bases uint8[n_reads*read_len] ASCII, read_off int64[n_reads+1])."""
    bases = np.empty(n_reads * read_len, np.uint8)
    lib().orc_synth_reads(C.c_uint64(seed), _p(genome), C.c_int64(genome_len), C.c_int64(first_read),
                          C.c_int64(n_reads), read_len, C.c_uint32(err_per_2_32), _p(bases))
    return bases, np.arange(n_reads + 1, dtype=np.int64) * read_len


# ---------------------------------------------------------------- f-4: contig RC de-duplication

def dedup_contigs(contigs, min_contig=500):
    """P/ReflexivDSDynamicKmerDedup.java assemblyFromKmer (:138-339) on a list of contig strings (ACGT), ids = positions ->
    dict(rounds=[list of strings after round 1, 2, 3], pairs=[..], candidates=[..], text=str)."""
    code = np.zeros(256, np.uint8)
    code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 3
    off = np.zeros(len(contigs) + 1, np.int64)
    off[1:] = np.cumsum([len(c) for c in contigs])
    bases = code[np.frombuffer("".join(contigs).encode(), np.uint8)] if off[-1] else np.zeros(0, np.uint8)
    n = len(contigs)
    cap_b = int(2 * off[-1] + 1024)
    bufs = [np.empty(cap_b, np.uint8) for _ in range(3)]
    offs = [np.empty(n + 2, np.int64) for _ in range(3)]
    rn, rb, rp, rc = (np.zeros(3, np.int64) for _ in range(4))
    m = lib().orc_dedup_contigs(_p(bases), _p(off), C.c_int64(n), _p(bufs[2]), C.c_int64(cap_b), _p(offs[2]), C.c_int64(n + 1),
                                _p(rn), _p(rb), _p(rp), _p(rc), _p(bufs[0]), _p(offs[0]), _p(bufs[1]), _p(offs[1]))
    if m < 0:
        raise ValueError("dedup output does not fit")
    nuc = np.frombuffer(b"ACGT", np.uint8)
    rounds = []
    for r in range(3):
        k = int(rn[r])
        rounds.append([bytes(nuc[bufs[r][offs[r][i]:offs[r][i + 1]]]).decode() for i in range(k)])
    ln = lib().orc_dedup_text(_p(bufs[2]), _p(offs[2]), C.c_int64(m), min_contig, None, C.c_int64(0))
    tb = np.empty(max(1, ln), np.uint8)
    lib().orc_dedup_text(_p(bufs[2]), _p(offs[2]), C.c_int64(m), min_contig, _p(tb), C.c_int64(ln))
    return dict(rounds=rounds, pairs=[int(x) for x in rp], candidates=[int(x) for x in rc], text=bytes(tb[:ln]).decode())


# ---------------------------------------------------------------- f-2: dynamic-k record format and passes

_CODE = np.full(256, 3, np.uint8)
_CODE[ord("A")], _CODE[ord("C")], _CODE[ord("G")] = 0, 1, 2
_NUC = np.frombuffer(b"ACGT", np.uint8)


@dataclass
class DynRecords:
    """the dynamic-k record set at sequence level: keys and extensions as base codes (0..3) with offsets"""
    key: np.ndarray        # uint8
    key_off: np.ndarray    # int64 [n+1]
    ext: np.ndarray        # uint8
    ext_off: np.ndarray    # int64 [n+1]
    marker: np.ndarray     # int32
    left: np.ndarray       # int32
    right: np.ndarray      # int32

    @property
    def n(self):
        return len(self.marker)

    def rows(self):
        """text rows as DSBinarySubKmerWith{Short,Long}ExtensionToString writes them"""
        out = []
        for i in range(self.n):
            k = bytes(_NUC[self.key[self.key_off[i]:self.key_off[i + 1]]]).decode()
            e = bytes(_NUC[self.ext[self.ext_off[i]:self.ext_off[i + 1]]]).decode()
            out.append((k, f"{int(self.marker[i])}|{int(self.left[i])}|{int(self.right[i])}", e))
        return out


def _dyn_pack(keys, exts, markers, lefts, rights):
    ko = np.zeros(len(keys) + 1, np.int64); ko[1:] = np.cumsum([len(x) for x in keys])
    eo = np.zeros(len(exts) + 1, np.int64); eo[1:] = np.cumsum([len(x) for x in exts])
    kb = _CODE[np.frombuffer("".join(keys).encode(), np.uint8)] if ko[-1] else np.zeros(0, np.uint8)
    eb = _CODE[np.frombuffer("".join(exts).encode(), np.uint8)] if eo[-1] else np.zeros(0, np.uint8)
    clamp = lambda v: max(-30000, min(30000, int(v)))
    return DynRecords(np.ascontiguousarray(kb), ko, np.ascontiguousarray(eb), eo, np.array(markers, np.int32),
                      np.array([clamp(v) for v in lefts], np.int32), np.array([clamp(v) for v in rights], np.int32))


def _attr(s):
    if s.endswith(")"):
        s = s[:-1]
    a = s.split("|")
    return int(a[0]), int(a[1]), int(a[2])


def dyn_binarize_kmers(rows):
    """DynamicKmerBinarizerFromReducedToSubKmer of FirstFour (:2931-3016): (k-mer text, "m|l|r") -> records: key = the
    k-mer without its last base, extension = that base, orientation forced to 1"""
    keys, exts, mk, lf, rt = [], [], [], [], []
    for kmer, attr in rows:
        if kmer.startswith("("):
            kmer = kmer[1:]
        _, l, r = _attr(attr)
        keys.append(kmer[:-1]); exts.append(kmer[-1]); mk.append(1); lf.append(l); rt.append(r)
    return _dyn_pack(keys, exts, mk, lf, rt)


def dyn_binarize_rows(rows):
    """DynamicKmerBinarizerFromReducedToSubKmer of Iteration: (sub-k-mer text, "m|l|r", extension text) -> records"""
    keys, exts, mk, lf, rt = [], [], [], [], []
    for k, attr, e in rows:
        if k.startswith("("):
            k = k[1:]
        m, l, r = _attr(attr)
        keys.append(k); exts.append(e); mk.append(m); lf.append(l); rt.append(r)
    return _dyn_pack(keys, exts, mk, lf, rt)


def dyn_gather(r: DynRecords, perm):
    keys = [r.key[r.key_off[i]:r.key_off[i + 1]] for i in perm]
    exts = [r.ext[r.ext_off[i]:r.ext_off[i + 1]] for i in perm]
    ko = np.zeros(len(perm) + 1, np.int64); ko[1:] = np.cumsum([len(x) for x in keys])
    eo = np.zeros(len(perm) + 1, np.int64); eo[1:] = np.cumsum([len(x) for x in exts])
    cat = lambda xs: np.ascontiguousarray(np.concatenate(xs)) if len(xs) and sum(len(x) for x in xs) else np.zeros(0, np.uint8)
    perm = np.asarray(perm, np.int64)
    return DynRecords(cat(keys), ko, cat(exts), eo, r.marker[perm].copy(), r.left[perm].copy(), r.right[perm].copy())


def dyn_sort(r: DynRecords):
    """sort("k-1") on the block form: element by element as signed longs, a proper prefix first; stable"""
    perm = np.empty(r.n, np.int64)
    lib().orc_dyn_sort_perm(_p(r.key), _p(r.key_off), C.c_int64(r.n), _p(perm))
    return dyn_gather(r, perm)


def dyn_partition_starts(r: DynRecords, P: int):
    """floor(p*n/P) moved forward past EQUAL keys (the order contract)"""
    n = r.n
    st, prev = [], 0
    key = lambda i: bytes(r.key[r.key_off[i]:r.key_off[i + 1]])
    for p in range(P):
        s = max(p * n // P, prev)
        while 0 < s < n and key(s) == key(s - 1):
            s += 1
        st.append(s); prev = s
    st.append(n)
    return np.array(st, np.int64)


def _dyn_out(n_cap, k_cap, e_cap):
    return (np.empty(max(1, k_cap), np.uint8), np.empty(n_cap + 1, np.int64), np.empty(max(1, e_cap), np.uint8), np.empty(n_cap + 1, np.int64),
            np.empty(max(1, n_cap), np.int32), np.empty(max(1, n_cap), np.int32), np.empty(max(1, n_cap), np.int32))


def _dyn_trim(o, m, nk, ne):
    ko = o[1][:m + 1].copy(); ko[m] = nk
    eo = o[3][:m + 1].copy(); eo[m] = ne
    return DynRecords(o[0][:nk].copy(), ko, o[2][:ne].copy(), eo, o[4][:m].copy(), o[5][:m].copy(), o[6][:m].copy())


def dyn_random_reflection(r: DynRecords, part_start):
    part_start = np.ascontiguousarray(part_start, np.int64)
    kc, ec = len(r.key) + len(r.ext), len(r.key) + len(r.ext)
    o = _dyn_out(r.n, kc, ec)
    m = lib().orc_dyn_random_reflection(_p(r.key), _p(r.key_off), _p(r.marker), _p(r.ext), _p(r.ext_off), _p(r.left), _p(r.right),
                                        C.c_int64(r.n), _p(part_start), len(part_start) - 1, _p(o[0]), C.c_int64(kc), _p(o[1]), _p(o[2]),
                                        C.c_int64(ec), _p(o[3]), _p(o[4]), _p(o[5]), _p(o[6]), C.c_int64(r.n))
    return _dyn_trim(o, m, int(r.key_off[-1]), int(r.ext_off[-1]))


def dyn_extend_pass(r: DynRecords, part_start, stage=0, start_iteration=5, start_marker=2):
    """one pass over rows sorted by key: stage 0 DSExtendReflexivKmer (FirstFour), 1 DSExtendReflexivKmerToArrayLoop"""
    part_start = np.ascontiguousarray(part_start, np.int64)
    P = len(part_start) - 1
    kc, ec = len(r.key) + len(r.ext) + 8, len(r.key) + len(r.ext) + 8
    o = _dyn_out(r.n, kc, ec)
    ops = np.empty(P + 1, np.int64)
    nk, ne = C.c_int64(0), C.c_int64(0)
    m = lib().orc_dyn_extend_pass(_p(r.key), _p(r.key_off), _p(r.marker), _p(r.ext), _p(r.ext_off), _p(r.left), _p(r.right), C.c_int64(r.n),
                                  _p(part_start), P, stage, start_iteration, start_marker, _p(o[0]), C.c_int64(kc), _p(o[1]), _p(o[2]),
                                  C.c_int64(ec), _p(o[3]), _p(o[4]), _p(o[5]), _p(o[6]), C.c_int64(r.n), _p(ops), C.byref(nk), C.byref(ne))
    assert nk.value <= kc and ne.value <= ec
    return _dyn_trim(o, m, int(nk.value), int(ne.value)), ops


def dyn_first_four(rows, P=1):
    """FirstFour.assemblyFromKmer (:137-224) on (k-mer text, "m|l|r") rows -> text rows; plus the rows after every operator"""
    trace = []
    r = dyn_binarize_kmers(rows)
    trace.append(("binarized", r.rows()))
    n = r.n
    st = np.array([p * n // P for p in range(P)] + [n], np.int64)
    r = dyn_random_reflection(r, st)
    trace.append(("random_reflection", r.rows()))
    for it in range(4):
        r = dyn_sort(r)
        r, _ = dyn_extend_pass(r, dyn_partition_starts(r, P), 0)
        trace.append((f"extend{it}", r.rows()))
    return r.rows(), trace


def dyn_iterations(rows, P=1, start=5, end=9):
    """Iteration.assemblyFromKmer (:134-205) on text rows -> text rows; plus the rows after every pass"""
    trace = []
    r = dyn_binarize_rows(rows)
    trace.append(("it_binarized", r.rows()))
    it = start
    while it <= end:
        it += 1
        r = dyn_sort(r)
        r, _ = dyn_extend_pass(r, dyn_partition_starts(r, P), 1, start)
        trace.append((f"it_extend{it}", r.rows()))
    return r.rows(), trace
