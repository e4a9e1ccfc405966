"""Host-side mirror of the reference's operator surface for the hot path, over the C ABI.

`Reflexiv` exposes one method per Spark operator class of P/ReflexivMain.java (names kept),
taking and returning flat numpy arrays in the reference's record layout, plus the resident
device pipeline (`count_reads_dev`, `assemble_dev`) that keeps everything in HBM.  Every
method runs the HIP kernels; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import time

import numpy as np

from . import _lib
from ._lib import Params, CRecords, RfxError, TWIN_DS, TWIN_RDD, RFX_OK, RFX_E_CAP  # noqa: F401


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@dataclass
class Records:
    """Flat SoA record set (include/reflexiv_hip.h rfx_records)."""
    key: np.ndarray
    marker: np.ndarray
    ext_off: np.ndarray
    ext: np.ndarray
    left: np.ndarray
    right: np.ndarray

    @property
    def n(self) -> int:
        return int(self.key.shape[0])

    @property
    def kw(self) -> int:
        """words per key: 1 (key is uint64[n]) or the second dimension of uint64[n, kw] (k > 32)"""
        return 1 if self.key.ndim == 1 else int(self.key.shape[1])

    def tuple_list(self):
        kk = [int(x) for x in self.key] if self.key.ndim == 1 else [tuple(int(x) for x in r) for r in self.key]
        return [(kk[i], int(self.marker[i]),
                 tuple(int(x) for x in self.ext[self.ext_off[i]:self.ext_off[i + 1]]),
                 int(self.left[i]), int(self.right[i])) for i in range(self.n)]

    def _c(self) -> CRecords:
        c = CRecords()
        c.n = self.n
        c.key, c.marker, c.ext_off = _p(self.key), _p(self.marker), _p(self.ext_off)
        c.ext, c.left, c.right = _p(self.ext), _p(self.left), _p(self.right)
        c.cap_n, c.cap_words = self.n, len(self.ext)
        c.key_words = self.kw
        return c

    @staticmethod
    def empty(cap_n: int, cap_words: int, kw: int = 1) -> "Records":
        cap_n, cap_words = max(1, cap_n), max(1, cap_words)
        key = np.empty(cap_n, np.uint64) if kw == 1 else np.empty((cap_n, kw), np.uint64)
        return Records(key, np.empty(cap_n, np.int32), np.empty(cap_n + 1, np.int64),
                       np.empty(cap_words, np.uint64), np.empty(cap_n, np.int32), np.empty(cap_n, np.int32))

    def _trim(self, c: CRecords) -> "Records":
        n = int(c.n)
        w = int(self.ext_off[n]) if n else 0
        return Records(self.key[:n].copy(), self.marker[:n].copy(), self.ext_off[:n + 1].copy(),
                       self.ext[:w].copy(), self.left[:n].copy(), self.right[:n].copy())


_DYN_CODE = np.full(256, 3, np.uint8)
_DYN_CODE[ord("A")], _DYN_CODE[ord("C")], _DYN_CODE[ord("G")] = 0, 1, 2      # nucleotideValue: A0 C1 G2, anything else 3
_DYN_NUC = np.frombuffer(b"ACGT", np.uint8)


@dataclass
class DynRecords:
    """The dynamic-k record set across the C ABI (rfx_dyn_records): keys and extensions as base codes with offsets."""
    key: np.ndarray        # uint8
    key_off: np.ndarray    # int64 [n+1]
    ext: np.ndarray        # uint8
    ext_off: np.ndarray    # int64 [n+1]
    marker: np.ndarray     # int32
    left: np.ndarray       # int32
    right: np.ndarray      # int32

    @property
    def n(self) -> int:
        return len(self.marker)

    def _c(self) -> "_lib.CDynRecords":
        c = _lib.CDynRecords()
        c.n = self.n
        c.key, c.key_off, c.ext, c.ext_off = (self.key.ctypes.data, self.key_off.ctypes.data, self.ext.ctypes.data, self.ext_off.ctypes.data)
        c.marker, c.left, c.right = self.marker.ctypes.data, self.left.ctypes.data, self.right.ctypes.data
        return c

    @staticmethod
    def empty(cap_n: int, cap_key: int, cap_ext: int) -> "DynRecords":
        return DynRecords(np.empty(max(1, cap_key), np.uint8), np.empty(cap_n + 1, np.int64), np.empty(max(1, cap_ext), np.uint8),
                          np.empty(cap_n + 1, np.int64), np.empty(max(1, cap_n), np.int32), np.empty(max(1, cap_n), np.int32),
                          np.empty(max(1, cap_n), np.int32))

    def _trim(self, c) -> "DynRecords":
        n = int(c.n)
        return DynRecords(self.key[:c.need_key].copy(), self.key_off[:n + 1].copy(), self.ext[:c.need_ext].copy(), self.ext_off[:n + 1].copy(),
                          self.marker[:n].copy(), self.left[:n].copy(), self.right[:n].copy())

    def rows(self):
        """the text rows DSBinarySubKmerWith{Short,Long}ExtensionToString write (FirstFour:226-263): (key, "m|l|r", extension)"""
        out = []
        for i in range(self.n):
            k = bytes(_DYN_NUC[self.key[self.key_off[i]:self.key_off[i + 1]]]).decode()
            e = bytes(_DYN_NUC[self.ext[self.ext_off[i]:self.ext_off[i + 1]]]).decode()
            out.append((k, f"{int(self.marker[i])}|{int(self.left[i])}|{int(self.right[i])}", e))
        return out

    @staticmethod
    def from_text(keys, exts, markers, lefts, rights) -> "DynRecords":
        ko = np.zeros(len(keys) + 1, np.int64); ko[1:] = np.cumsum([len(x) for x in keys])
        eo = np.zeros(len(exts) + 1, np.int64); eo[1:] = np.cumsum([len(x) for x in exts])
        kb = _DYN_CODE[np.frombuffer("".join(keys).encode(), np.uint8)] if ko[-1] else np.zeros(1, np.uint8)
        eb = _DYN_CODE[np.frombuffer("".join(exts).encode(), np.uint8)] if eo[-1] else np.zeros(1, np.uint8)
        cl = lambda v: max(-30000, min(30000, int(v)))             # buildingAlongFromThreeInt read back (FirstFour:2340-2366)
        return DynRecords(np.ascontiguousarray(kb), ko, np.ascontiguousarray(eb), eo, np.array(markers, np.int32),
                          np.array([cl(v) for v in lefts], np.int32), np.array([cl(v) for v in rights], np.int32))

    @staticmethod
    def from_kmer_rows(rows) -> "DynRecords":
        """DynamicKmerBinarizerFromReducedToSubKmer of FirstFour (:2931-3016): (k-mer text, "m|l|r") rows"""
        keys, exts, mk, lf, rt = [], [], [], [], []
        for kmer, attr in rows:
            kmer = kmer[1:] if kmer.startswith("(") else kmer
            a = (attr[:-1] if attr.endswith(")") else attr).split("|")
            keys.append(kmer[:-1]); exts.append(kmer[-1]); mk.append(1); lf.append(int(a[1])); rt.append(int(a[2]))
        return DynRecords.from_text(keys, exts, mk, lf, rt)

    @staticmethod
    def from_rows(rows) -> "DynRecords":
        """DynamicKmerBinarizerFromReducedToSubKmer of Iteration: (sub-k-mer text, "m|l|r", extension text) rows"""
        keys, exts, mk, lf, rt = [], [], [], [], []
        for k, attr, e in rows:
            k = k[1:] if k.startswith("(") else k
            a = (attr[:-1] if attr.endswith(")") else attr).split("|")
            keys.append(k); exts.append(e); mk.append(int(a[0])); lf.append(int(a[1])); rt.append(int(a[2]))
        return DynRecords.from_text(keys, exts, mk, lf, rt)


def as_records(r) -> Records:
    """Accept any object with key/marker/ext_off/ext/left/right arrays (e.g. the oracle's Records)."""
    return Records(np.ascontiguousarray(r.key, np.uint64), np.ascontiguousarray(r.marker, np.int32),
                   np.ascontiguousarray(r.ext_off, np.int64), np.ascontiguousarray(r.ext, np.uint64),
                   np.ascontiguousarray(r.left, np.int32), np.ascontiguousarray(r.right, np.int32))


def sub_words(k: int) -> int:
    """words of a (k-1)-mer key (subKmerBinarySlots, U/DefaultParam.java:94): 1 up to k = 32"""
    return 1 if k <= 32 else (k - 2) // 31 + 1


def asm_words(k: int) -> int:
    """words of a k-mer in the assembler's 31-bases-per-word layout (kmerBinarySlotsAssemble, :85)"""
    return 1 if k <= 31 else (k - 1) // 31 + 1


def default_params(**kw) -> Params:
    p = Params()
    _lib.lib().rfx_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Reflexiv:
    """One context = one MI355X + one HIP stream."""

    def __init__(self, device: int = -1):
        self.L = _lib.lib()
        ctx = C.c_void_p()
        st = self.L.rfx_ctx_create(device, C.byref(ctx))
        if st != RFX_OK:
            raise RfxError(st, "rfx_ctx_create", "a gfx950 (MI355X) GPU is required; there is no CPU fallback")
        self.ctx = ctx

    def close(self):
        if getattr(self, "comm", None) and getattr(self, "ctx", None):
            self.comm_destroy()
        if getattr(self, "ctx", None):
            self.L.rfx_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st, where):
        if st != RFX_OK:
            raise RfxError(st, where, (self.L.rfx_last_error(self.ctx) or b"").decode())

    def use_stream(self, hip_stream: int):
        self._check(self.L.rfx_ctx_set_stream(self.ctx, C.c_void_p(hip_stream)), "rfx_ctx_set_stream")

    def sync(self):
        self._check(self.L.rfx_ctx_sync(self.ctx), "rfx_ctx_sync")

    # ------------------------------------------------ operators on host arrays

    def ReverseComplementKmerBinaryExtraction(self, bases, read_off, k=31, front_clip=0, end_clip=0):
        """P/ReflexivMain.java:3013-3075 -> canonical k-mers in read/window order."""
        bases = np.ascontiguousarray(bases, np.uint8)
        read_off = np.ascontiguousarray(read_off, np.int64)
        nr = len(read_off) - 1
        n = C.c_int64(0)
        st = self.L.rfx_extract_canon(self.ctx, _p(bases), _p(read_off), C.c_int64(nr), k, front_clip, end_clip,
                                      None, C.c_int64(0), C.byref(n))
        if st not in (RFX_OK, RFX_E_CAP):
            self._check(st, "rfx_extract_canon")
        out = np.empty(max(1, n.value), np.uint64)
        self._check(self.L.rfx_extract_canon(self.ctx, _p(bases), _p(read_off), C.c_int64(nr), k, front_clip,
                                             end_clip, _p(out), C.c_int64(n.value), C.byref(n)),
                    "rfx_extract_canon")
        return out[:n.value]

    def KmerCounting_and_CoverageFilter(self, kmers, min_cov=2, max_cov=10_000_000, twin=TWIN_DS):
        """reduceByKey(KmerCounting) + filter(KmerCoverageFilter), P/ReflexivMain.java:155-163."""
        kmers = np.ascontiguousarray(kmers, np.uint64)
        n = len(kmers)
        keys = np.empty(max(1, n), np.uint64)
        counts = np.empty(max(1, n), np.int32)
        m, d = C.c_int64(0), C.c_int64(0)
        self._check(self.L.rfx_count_filter(self.ctx, _p(kmers), C.c_int64(n), min_cov, max_cov, twin, _p(keys),
                                            _p(counts), C.c_int64(n), C.byref(m), C.byref(d)), "rfx_count_filter")
        return keys[:m.value].copy(), counts[:m.value].copy(), int(d.value)

    # k > 31: the counter's W = k//32+1 word k-mers (P/ReflexivDataFrameCounter64.java)
    def ReverseComplementKmerBinaryExtractionFromDataset64(self, bases, read_off, k=63, front_clip=0, end_clip=0):
        """:401-650 -> canonical k-mers uint64[n, W] in read/window order."""
        bases = np.ascontiguousarray(bases, np.uint8)
        read_off = np.ascontiguousarray(read_off, np.int64)
        nr = len(read_off) - 1
        W = k // 32 + 1
        n = C.c_int64(0)
        st = self.L.rfx_extract_canon_w(self.ctx, _p(bases), _p(read_off), C.c_int64(nr), k, front_clip, end_clip,
                                        None, C.c_int64(0), C.byref(n))
        if st not in (RFX_OK, RFX_E_CAP):
            self._check(st, "rfx_extract_canon_w")
        out = np.empty((max(1, n.value), W), np.uint64)
        self._check(self.L.rfx_extract_canon_w(self.ctx, _p(bases), _p(read_off), C.c_int64(nr), k, front_clip,
                                               end_clip, _p(out), C.c_int64(n.value), C.byref(n)),
                    "rfx_extract_canon_w")
        return out[:n.value]

    def groupBy_count_filter_w(self, kmers, k=63, min_cov=2, max_cov=10_000_000):
        """groupBy("kmerBlocks").count() + the two filters, :191-205 -> (keys[m, W], counts int64[m], n_distinct)."""
        W = k // 32 + 1
        kmers = np.ascontiguousarray(kmers, np.uint64).reshape(-1, W)
        n = len(kmers)
        keys = np.empty((max(1, n), W), np.uint64)
        counts = np.empty(max(1, n), np.int64)
        m, d = C.c_int64(0), C.c_int64(0)
        self._check(self.L.rfx_count_filter_w(self.ctx, _p(kmers), C.c_int64(n), k, min_cov, max_cov, _p(keys),
                                              _p(counts), C.c_int64(n), C.byref(m), C.byref(d)), "rfx_count_filter_w")
        return keys[:m.value].copy(), counts[:m.value].copy(), int(d.value)

    def kmers_per_read_w(self, read_len, k, front_clip=0, end_clip=0) -> int:
        return int(self.L.rfx_kmers_per_read_w(read_len, k, front_clip, end_clip))

    def count_reads_w_dev(self, d_words: int, n_reads: int, words_per_read: int, read_len: int, k: int,
                          d_out_keys: int, d_out_counts: int, cap: int, min_cov=2, max_cov=10_000_000,
                          front_clip=0, end_clip=0):
        """k > 31 twin of count_reads_dev -> (n_survivors, n_distinct, n_instances); keys cap*W words, counts int64."""
        n, d, inst = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        st = self.L.rfx_dev_count_reads_w(self.ctx, C.c_void_p(d_words), C.c_int64(n_reads), words_per_read, read_len,
                                          k, front_clip, end_clip, min_cov, max_cov, C.c_void_p(d_out_keys),
                                          C.c_void_p(d_out_counts), C.c_int64(cap), C.byref(n), C.byref(d),
                                          C.byref(inst))
        if st == RFX_E_CAP:
            raise RfxError(st, "rfx_dev_count_reads_w", f"needs room for {n.value} survivors, cap is {cap}")
        self._check(st, "rfx_dev_count_reads_w")
        return int(n.value), int(d.value), int(inst.value)

    def KmerReverseComplement_and_ForwardSubKmerExtraction(self, keys, counts, k=31) -> Records:
        """k > 31 (DSKmerReverseComplement + DSForwardSubKmerExtraction of ReflexivDSMain64): keys uint64[n, (k-1)//31+1]
        in the assembler layout."""
        keys = np.ascontiguousarray(keys, np.uint64)
        counts = np.ascontiguousarray(counts, np.int32)
        n = len(counts)
        assert keys.size == n * asm_words(k), (keys.shape, k)
        out = Records.empty(2 * n, 2 * n, sub_words(k))
        c = out._c(); c.cap_n = 2 * n; c.cap_words = 2 * n
        self._check(self.L.rfx_rc_expand_subkmer(self.ctx, _p(keys), _p(counts), C.c_int64(n), k, C.byref(c)),
                    "rfx_rc_expand_subkmer")
        return out._trim(c)

    def sortByKey(self, r, P: int):
        """-> (sorted Records, part_start[P+1])."""
        r = as_records(r)
        out = Records.empty(r.n, len(r.ext), r.kw)
        ci, co = r._c(), out._c()
        co.cap_n, co.cap_words = r.n, len(r.ext)
        ps = np.empty(P + 1, np.int64)
        self._check(self.L.rfx_sort_records(self.ctx, C.byref(ci), P, C.byref(co), _p(ps)), "rfx_sort_records")
        return out._trim(co), ps

    def _fork(self, fn, name, r, part_start, k, min_error_cov, twin):
        r = as_records(r)
        part_start = np.ascontiguousarray(part_start, np.int64)
        P = len(part_start) - 1
        out = Records.empty(r.n, r.n, r.kw)
        ci, co = r._c(), out._c()
        co.cap_n, co.cap_words = r.n, r.n
        ops = np.empty(P + 1, np.int64)
        self._check(fn(self.ctx, C.byref(ci), _p(part_start), P, k, min_error_cov, twin, C.byref(co), _p(ops)), name)
        return out._trim(co), ops

    def FilterForkSubKmer(self, r, part_start, k=31, min_error_cov=8, twin=TWIN_DS):
        return self._fork(self.L.rfx_fork_filter_forward, "rfx_fork_filter_forward", r, part_start, k,
                          min_error_cov, twin)

    def FilterForkReflectedSubKmer(self, r, part_start, k=31, min_error_cov=8, twin=TWIN_DS):
        return self._fork(self.L.rfx_fork_filter_reflected, "rfx_fork_filter_reflected", r, part_start, k,
                          min_error_cov, twin)

    def ReflectedSubKmerExtractionFromForward(self, r, k=31) -> Records:
        r = as_records(r)
        out = Records.empty(r.n, r.n, r.kw)
        ci, co = r._c(), out._c()
        co.cap_n, co.cap_words = r.n, r.n
        self._check(self.L.rfx_reflect_from_forward(self.ctx, C.byref(ci), k, C.byref(co)),
                    "rfx_reflect_from_forward")
        return out._trim(co)

    def kmerRandomReflection(self, r, part_start, k=31) -> Records:
        r = as_records(r)
        part_start = np.ascontiguousarray(part_start, np.int64)
        out = Records.empty(r.n, r.n, r.kw)
        ci, co = r._c(), out._c()
        co.cap_n, co.cap_words = r.n, r.n
        self._check(self.L.rfx_random_reflection(self.ctx, C.byref(ci), _p(part_start), len(part_start) - 1, k,
                                                 C.byref(co)), "rfx_random_reflection")
        return out._trim(co)

    def ExtendReflexivKmer(self, r, part_start, k=31, twin=TWIN_DS, stage=2, scramble=2):
        """One extend pass (stage 0: ExtendReflexivKmer, 1: ...ToArrayFirstTime, 2: ...ToArrayLoop).
        scramble = 3: DSExtendReflexivKmerToArrayLoop of ReflexivDSMain64 after param.scramble went to 3
        (the task's emission marker starts at 1, :7484-7486)."""
        r = as_records(r)
        part_start = np.ascontiguousarray(part_start, np.int64)
        P = len(part_start) - 1
        out = Records.empty(r.n, len(r.ext), r.kw)
        ci, co = r._c(), out._c()
        co.cap_n, co.cap_words = r.n, len(r.ext)
        ops = np.empty(P + 1, np.int64)
        if scramble != 2:
            self._check(self.L.rfx_extend_pass_w(self.ctx, C.byref(ci), _p(part_start), P, k, stage, scramble,
                                                 C.byref(co), _p(ops)), "rfx_extend_pass_w")
        else:
            self._check(self.L.rfx_extend_pass(self.ctx, C.byref(ci), _p(part_start), P, k, twin, stage, C.byref(co),
                                               _p(ops)), "rfx_extend_pass")
        return out._trim(co), ops

    def extras_operator(self, op: int, r, part_start, k: int):
        """One operator class of the k > 31 from-counts extras (P/ReflexivDSMain64.java:584-619, 672-712): op 0
        DSReflexivAndForwardKmer, 1 DSFilterExtendableKmerPairs, 2 DSFilterUnExtendableKmer, 3 DSFilterStillExtendableKmerFromPairs,
        4 DSFilterStillExtendableKmerEnds, 5 / 6 DSFilterUnExtendableKmerLeftEnds / ...RightEnds -> (Records, out_part_start)"""
        r = as_records(r)
        part_start = np.ascontiguousarray(part_start, np.int64)
        P = len(part_start) - 1
        mul = 2 if op == 0 else 1
        out = Records.empty(mul * r.n, mul * len(r.ext), r.kw)
        ci, co = r._c(), out._c()
        co.cap_n, co.cap_words = mul * r.n, mul * len(r.ext)
        ops = np.empty(P + 1, np.int64)
        self._check(self.L.rfx_extras_operator(self.ctx, op, C.byref(ci), _p(part_start), P, k, C.byref(co), _p(ops)),
                    "rfx_extras_operator")
        return out._trim(co), ops

    def KmerToContig(self, r, k=31, min_contig=500, twin=TWIN_DS):
        r = as_records(r)
        ci = r._c()
        ln, nc = C.c_int64(0), C.c_int64(0)
        st = self.L.rfx_contigs_text(self.ctx, C.byref(ci), k, min_contig, twin, None, C.c_int64(0), C.byref(ln),
                                     C.byref(nc))
        if st not in (RFX_OK, RFX_E_CAP):
            self._check(st, "rfx_contigs_text")
        buf = np.empty(max(1, ln.value), np.uint8)
        self._check(self.L.rfx_contigs_text(self.ctx, C.byref(ci), k, min_contig, twin, _p(buf),
                                            C.c_int64(ln.value), C.byref(ln), C.byref(nc)), "rfx_contigs_text")
        return bytes(buf[:ln.value]).decode(), int(nc.value)

    # ------------------------------------------------ resident device pipeline

    def encode_reads_dev(self, d_bases: int, d_read_off: int, n_reads: int, words_per_read: int, d_words: int,
                         d_read_len: int = 0):
        self._check(self.L.rfx_dev_encode_reads(self.ctx, C.c_void_p(d_bases), C.c_void_p(d_read_off),
                                                C.c_int64(n_reads), words_per_read, C.c_void_p(d_words),
                                                C.c_void_p(d_read_len) if d_read_len else None),
                    "rfx_dev_encode_reads")

    def kmers_per_read(self, read_len, k, front_clip=0, end_clip=0) -> int:
        return int(self.L.rfx_kmers_per_read(read_len, k, front_clip, end_clip))

    def count_reads_dev(self, d_words: int, n_reads: int, words_per_read: int, read_len: int, k: int,
                        d_out_keys: int, d_out_counts: int, cap: int, min_cov=2, max_cov=10_000_000,
                        twin=TWIN_DS, front_clip=0, end_clip=0):
        """extract + reduceByKey + filter from packed reads in HBM -> (n_survivors, n_distinct, n_instances)."""
        n, d, inst = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        st = self.L.rfx_dev_count_reads(self.ctx, C.c_void_p(d_words), C.c_int64(n_reads), words_per_read, read_len,
                                        k, front_clip, end_clip, min_cov, max_cov, twin, None, C.c_int64(0),
                                        C.c_void_p(d_out_keys), C.c_void_p(d_out_counts), C.c_int64(cap),
                                        C.byref(n), C.byref(d), C.byref(inst))
        if st == RFX_E_CAP:
            raise RfxError(st, "rfx_dev_count_reads", f"needs room for {n.value} survivors, cap is {cap}")
        self._check(st, "rfx_dev_count_reads")
        return int(n.value), int(d.value), int(inst.value)

    def count_reads_ragged_dev(self, d_words: int, d_read_len: int, n_reads: int, words_per_read: int, max_read_len: int,
                               k: int, d_out_keys: int, d_out_counts: int, cap: int, min_cov=2, max_cov=10_000_000,
                               twin=TWIN_DS, front_clip=0, end_clip=0):
        """count_reads_dev for reads of different lengths (d_read_len: uint32 per read, as encode_reads_dev writes)."""
        n, d, inst = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        st = self.L.rfx_dev_count_reads_ragged(self.ctx, C.c_void_p(d_words), C.c_void_p(d_read_len), C.c_int64(n_reads),
                                               words_per_read, max_read_len, k, front_clip, end_clip, min_cov, max_cov,
                                               twin, C.c_void_p(d_out_keys), C.c_void_p(d_out_counts), C.c_int64(cap),
                                               C.byref(n), C.byref(d), C.byref(inst))
        if st == RFX_E_CAP:
            raise RfxError(st, "rfx_dev_count_reads_ragged", f"needs room for {n.value} survivors, cap is {cap}")
        self._check(st, "rfx_dev_count_reads_ragged")
        return int(n.value), int(d.value), int(inst.value)

    def bucket_wide_by_owner_dev(self, d_words: int, n_reads: int, words_per_read: int, read_len: int, k: int,
                                 n_owners: int, d_out: int, cap_elems: int, d_owner_off: int, front_clip=0, end_clip=0):
        """k = 33..63: two-word k-mers (16 B each) grouped by owning rank -> owner_off[n_owners+1] (host)."""
        h = np.empty(n_owners + 1, np.int64)
        self._check(self.L.rfx_dev_bucket_wide_by_owner(self.ctx, C.c_void_p(d_words), C.c_int64(n_reads), words_per_read,
                                                        read_len, k, front_clip, end_clip, n_owners, C.c_void_p(d_out),
                                                        C.c_int64(cap_elems), C.c_void_p(d_owner_off), _p(h)),
                    "rfx_dev_bucket_wide_by_owner")
        return h

    def count_wide_elems_dev(self, d_elems: int, n_elems: int, k: int, d_out_keys: int, d_out_counts: int, cap: int,
                             min_cov=2, max_cov=10_000_000):
        m, d = C.c_int64(0), C.c_int64(0)
        st = self.L.rfx_dev_count_wide_elems(self.ctx, C.c_void_p(d_elems), C.c_int64(n_elems), k, min_cov, max_cov,
                                             C.c_void_p(d_out_keys), C.c_void_p(d_out_counts), C.c_int64(cap),
                                             C.byref(m), C.byref(d))
        if st == RFX_E_CAP:
            raise RfxError(st, "rfx_dev_count_wide_elems", f"needs room for {m.value} survivors, cap is {cap}")
        self._check(st, "rfx_dev_count_wide_elems")
        return int(m.value), int(d.value)

    def count_kmers_dev(self, d_kmers: int, n: int, d_out_keys: int, d_out_counts: int, cap: int, min_cov=2,
                        max_cov=10_000_000, twin=TWIN_DS):
        m, d = C.c_int64(0), C.c_int64(0)
        st = self.L.rfx_dev_count_kmers(self.ctx, C.c_void_p(d_kmers), C.c_int64(n), min_cov, max_cov, twin, None,
                                        C.c_int64(0), C.c_void_p(d_out_keys), C.c_void_p(d_out_counts),
                                        C.c_int64(cap), C.byref(m), C.byref(d))
        if st == RFX_E_CAP:
            raise RfxError(st, "rfx_dev_count_kmers", f"needs room for {m.value} survivors, cap is {cap}")
        self._check(st, "rfx_dev_count_kmers")
        return int(m.value), int(d.value)

    def bucket_by_owner_dev(self, d_words: int, n_reads: int, words_per_read: int, read_len: int, k: int,
                            n_owners: int, d_out: int, cap: int, d_owner_off: int, front_clip=0, end_clip=0):
        h = np.empty(n_owners + 1, np.int64)
        self._check(self.L.rfx_dev_bucket_by_owner(self.ctx, C.c_void_p(d_words), C.c_int64(n_reads), words_per_read,
                                                   read_len, k, front_clip, end_clip, n_owners, C.c_void_p(d_out),
                                                   C.c_int64(cap), C.c_void_p(d_owner_off), _p(h)),
                    "rfx_dev_bucket_by_owner")
        return h

    def bucket_records_by_owner_dev(self, d_words: int, n_reads: int, words_per_read: int, read_len: int, k: int,
                                    n_owners: int, d_out: int, cap_records: int, d_owner_off: int,
                                    front_clip=0, end_clip=0):
        """Super-k-mer records (16 B each) grouped by owning rank.  d_out = 0: only returns how many
        records there are.  -> (n_records, owner_off[n_owners+1] or None)"""
        h = np.empty(n_owners + 1, np.int64)
        nrec = C.c_int64(0)
        st = self.L.rfx_dev_bucket_records_by_owner(self.ctx, C.c_void_p(d_words), C.c_int64(n_reads), words_per_read,
                                                    read_len, k, front_clip, end_clip, n_owners,
                                                    C.c_void_p(d_out) if d_out else None, C.c_int64(cap_records),
                                                    C.c_void_p(d_owner_off), _p(h), C.byref(nrec))
        if st == RFX_E_CAP:
            return int(nrec.value), None
        self._check(st, "rfx_dev_bucket_records_by_owner")
        return int(nrec.value), h

    def count_records_dev(self, d_records: int, n_records: int, n_instances_hint: int, k: int, d_out_keys: int,
                          d_out_counts: int, cap: int, min_cov=2, max_cov=10_000_000, twin=TWIN_DS):
        m, d = C.c_int64(0), C.c_int64(0)
        st = self.L.rfx_dev_count_records(self.ctx, C.c_void_p(d_records), C.c_int64(n_records),
                                          C.c_int64(n_instances_hint), k, min_cov, max_cov, twin,
                                          C.c_void_p(d_out_keys), C.c_void_p(d_out_counts), C.c_int64(cap),
                                          C.byref(m), C.byref(d))
        if st == RFX_E_CAP:
            raise RfxError(st, "rfx_dev_count_records", f"needs room for {m.value} survivors, cap is {cap}")
        self._check(st, "rfx_dev_count_records")
        return int(m.value), int(d.value)

    def bucket_wide_records_by_owner_dev(self, d_words: int, n_reads: int, wpr: int, read_len: int, k: int, n_owners: int,
                                         d_out_records: int, cap_records: int, d_owner_off: int, front_clip=0, end_clip=0):
        """k = 33..63: 32-byte super-k-mer records grouped by owner -> (n_records, owner_off host or None when the
        buffer is missing / short: n_records is then the capacity needed)."""
        nrec = C.c_int64(0)
        h = np.zeros(n_owners + 1, np.int64)
        st = self.L.rfx_dev_bucket_wide_records_by_owner(self.ctx, C.c_void_p(d_words), C.c_int64(n_reads), wpr, read_len, k,
                                                         front_clip, end_clip, n_owners, C.c_void_p(d_out_records),
                                                         C.c_int64(cap_records), C.c_void_p(d_owner_off), _p(h), C.byref(nrec))
        if st == RFX_E_CAP:
            return int(nrec.value), None
        self._check(st, "rfx_dev_bucket_wide_records_by_owner")
        return int(nrec.value), h

    def count_wide_records_dev(self, d_records: int, n_records: int, n_instances_hint: int, k: int, d_out_keys: int,
                               d_out_counts: int, cap: int, min_cov=2, max_cov=10_000_000):
        m, d = C.c_int64(0), C.c_int64(0)
        st = self.L.rfx_dev_count_wide_records(self.ctx, C.c_void_p(d_records), C.c_int64(n_records),
                                               C.c_int64(n_instances_hint), k, min_cov, max_cov, C.c_void_p(d_out_keys),
                                               C.c_void_p(d_out_counts), C.c_int64(cap), C.byref(m), C.byref(d))
        if st == RFX_E_CAP:
            raise RfxError(st, "rfx_dev_count_wide_records", f"needs room for {m.value} survivors, cap is {cap}")
        self._check(st, "rfx_dev_count_wide_records")
        return int(m.value), int(d.value)

    # ---- the exchange after a local combine (reduceByKey's map-side combine, P/ReflexivMain.java:155)
    def combine_reads_dev(self, d_words: int, n_reads: int, wpr: int, read_len: int, k: int, n_owners: int,
                          d_scratch_pairs: int, d_out_pairs: int, cap_pairs: int, d_owner_off: int,
                          front_clip=0, end_clip=0):
        """reads -> every distinct canonical k-mer with its local count as 16-byte {k-mer, count} pairs grouped
        by owner (no filter) -> (n_pairs, owner_off host int64[n_owners+1], instances); RfxError(RFX_E_CAP)
        carries the capacity needed in `.need`."""
        m, inst = C.c_int64(0), C.c_int64(0)
        h = np.zeros(n_owners + 1, np.int64)
        st = self.L.rfx_dev_combine_reads(self.ctx, C.c_void_p(d_words), C.c_int64(n_reads), wpr, read_len, k,
                                          front_clip, end_clip, n_owners, C.c_void_p(d_scratch_pairs),
                                          C.c_void_p(d_out_pairs), C.c_int64(cap_pairs), C.c_void_p(d_owner_off), _p(h),
                                          C.byref(m), C.byref(inst))
        if st == RFX_E_CAP:
            e = RfxError(st, "rfx_dev_combine_reads", f"needs room for {m.value} pairs, cap is {cap_pairs}")
            e.need = int(m.value)
            raise e
        self._check(st, "rfx_dev_combine_reads")
        return int(m.value), h, int(inst.value)

    def bucket_pairs_by_owner_dev(self, d_pairs: int, n_pairs: int, n_owners: int, d_out_pairs: int, d_owner_off: int):
        h = np.zeros(n_owners + 1, np.int64)
        self._check(self.L.rfx_dev_bucket_pairs_by_owner(self.ctx, C.c_void_p(d_pairs), C.c_int64(n_pairs), n_owners,
                                                         C.c_void_p(d_out_pairs), C.c_void_p(d_owner_off), _p(h)),
                    "rfx_dev_bucket_pairs_by_owner")
        return h

    def merge_pairs_dev(self, d_pairs: int, n_pairs: int, k: int, d_out_keys: int, d_out_counts: int, cap: int,
                        min_cov=2, max_cov=10_000_000, twin=TWIN_DS):
        m, d = C.c_int64(0), C.c_int64(0)
        st = self.L.rfx_dev_merge_pairs(self.ctx, C.c_void_p(d_pairs), C.c_int64(n_pairs), k, min_cov, max_cov, twin,
                                        C.c_void_p(d_out_keys), C.c_void_p(d_out_counts), C.c_int64(cap),
                                        C.byref(m), C.byref(d))
        if st == RFX_E_CAP:
            raise RfxError(st, "rfx_dev_merge_pairs", f"needs room for {m.value} survivors, cap is {cap}")
        self._check(st, "rfx_dev_merge_pairs")
        return int(m.value), int(d.value)

    def _text_buffer(self, cap: int):
        """The contig text of the device-resident drivers lands in one host buffer that the object keeps (grow-only):
        a fresh multi-hundred-megabyte mapping per call costs its page faults on every megabyte the text touches."""
        buf = getattr(self, "_textbuf", None)
        if buf is None or len(buf) < cap:
            buf = self._textbuf = np.empty(cap, np.uint8)
        return buf, len(buf)

    def assemble_dev(self, d_keys: int, d_counts: int, n: int, prm: Params):
        """Driver P/ReflexivMain.java:168-310 from the filtered (kmer,count) list in HBM
        -> (contig text, n_contigs, trace)."""
        trace = np.zeros(prm.max_iter + 8, np.int64)
        ln, nc, ntr = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        buf, cap = self._text_buffer(4 * (n + 16) * (prm.k + 8) + 1024)
        self._check(self.L.rfx_dev_assemble(self.ctx, C.c_void_p(d_keys), C.c_void_p(d_counts), C.c_int64(n),
                                            C.byref(prm), _p(buf), C.c_int64(cap), C.byref(ln), C.byref(nc),
                                            _p(trace), C.c_int64(len(trace)), C.byref(ntr)), "rfx_dev_assemble")
        return str(memoryview(buf)[:ln.value], "ascii"), int(nc.value), [int(x) for x in trace[:ntr.value]]

    def order_kmers_w_dev(self, d_keys: int, d_counts: int, n: int, k: int):
        """k = 33..63: (two-word k-mer, int64 count) pairs in any order -> ascending k-mer order, in place."""
        self._check(self.L.rfx_dev_order_kmers_w(self.ctx, C.c_void_p(d_keys), C.c_void_p(d_counts), C.c_int64(n), k),
                    "rfx_dev_order_kmers_w")

    def counter_to_asm_dev(self, d_keys32: int, d_counts64: int, n: int, k: int, d_out_kmers: int, d_out_counts: int,
                           min_cov=2, max_cov=10_000_000) -> int:
        """KmerBinarizer + the from-counts filter (P/ReflexivDSMain64.java:10772-10836, :473-478) in HBM: the counter's
        (k//32+1)-word k-mers and int64 counts -> (k-1)//31+1 words of 31 bases and int32 counts -> kept."""
        m = C.c_int64(0)
        self._check(self.L.rfx_dev_counter_to_asm(self.ctx, C.c_void_p(d_keys32), C.c_void_p(d_counts64), C.c_int64(n), k,
                                                  min_cov, max_cov, C.c_void_p(d_out_kmers), C.c_void_p(d_out_counts),
                                                  C.byref(m)), "rfx_dev_counter_to_asm")
        return int(m.value)

    def assemble_w_dev(self, d_kmers: int, d_counts: int, n: int, prm: Params):
        """k > 31 driver ReflexivDSMain64.assemblyFromKmer (:458-826, without the extras of :584-619 / :672-712)
        from the filtered (k-mer, count) list in HBM (assembler layout) -> (contig text, n_contigs, trace)."""
        trace = np.zeros(prm.max_iter + 8, np.int64)
        ln, nc, ntr = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        buf, cap = self._text_buffer(4 * (n + 16) * (prm.k + 8) + 1024)
        self._check(self.L.rfx_dev_assemble_w(self.ctx, C.c_void_p(d_kmers), C.c_void_p(d_counts), C.c_int64(n),
                                              C.byref(prm), _p(buf), C.c_int64(cap), C.byref(ln), C.byref(nc),
                                              _p(trace), C.c_int64(len(trace)), C.byref(ntr)), "rfx_dev_assemble_w")
        return str(memoryview(buf)[:ln.value], "ascii"), int(nc.value), [int(x) for x in trace[:ntr.value]]

    def synth_genome_dev(self, seed: int, genome_len: int, d_genome: int):
        self._check(self.L.rfx_dev_synth_genome(self.ctx, C.c_uint64(seed), C.c_int64(genome_len),
                                                C.c_void_p(d_genome)), "rfx_dev_synth_genome")

    def synth_reads_dev(self, seed: int, d_genome: int, genome_len: int, first_read: int, n_reads: int,
                        read_len: int, words_per_read: int, d_words: int, err_per_2_32: int = 21474836):
        self._check(self.L.rfx_dev_synth_reads(self.ctx, C.c_uint64(seed), C.c_void_p(d_genome),
                                               C.c_int64(genome_len), C.c_int64(first_read), C.c_int64(n_reads),
                                               read_len, C.c_uint32(err_per_2_32), words_per_read,
                                               C.c_void_p(d_words)), "rfx_dev_synth_reads")

    def sort_pairs_dev(self, d_keys: int, d_vals: int, n: int, key_bits: int, d_tmp_keys: int, d_tmp_vals: int):
        self._check(self.L.rfx_dev_sort_pairs(self.ctx, C.c_void_p(d_keys), C.c_void_p(d_vals), C.c_int64(n),
                                              key_bits, C.c_void_p(d_tmp_keys), C.c_void_p(d_tmp_vals)),
                    "rfx_dev_sort_pairs")

    def assemble_reads_ptr(self, bases_ptr: int, n_bases: int, read_off, prm: Params):
        """rfx_assemble_reads: ASCII reads in HOST memory (any lengths; bases_ptr may be pinned memory) -> upload, 2-bit
        encode, count / filter, the driver -> (text, n_contigs, trace, kept).  k <= 31."""
        read_off = np.ascontiguousarray(read_off, np.int64)
        n_reads = len(read_off) - 1
        trace = np.zeros(prm.max_iter + 8, np.int64)
        ln, nc, ntr, kept = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        buf, cap = self._text_buffer(1 << 24)
        while True:
            st = self.L.rfx_assemble_reads(self.ctx, C.c_void_p(bases_ptr), _p(read_off), C.c_int64(n_reads), C.byref(prm), _p(buf),
                                           C.c_int64(cap), C.byref(ln), C.byref(nc), _p(trace), C.c_int64(len(trace)), C.byref(ntr),
                                           C.byref(kept))
            if st == RFX_E_CAP and ln.value > cap:
                buf, cap = self._text_buffer(int(ln.value))
                continue
            self._check(st, "rfx_assemble_reads")
            return str(memoryview(buf)[:ln.value], "ascii"), int(nc.value), [int(x) for x in trace[:ntr.value]], int(kept.value)

    def assemble_reads(self, bases, read_off, prm: Params):
        bases = np.ascontiguousarray(bases, np.uint8)
        return self.assemble_reads_ptr(bases.ctypes.data, len(bases), read_off, prm)

    # ------------------------------------------------ f-2: the dynamic-k record format and passes
    def _dyn_call(self, fn, name, r: DynRecords, make_args, grow=1):
        cap_n = r.n * grow
        cap_b = (len(r.key) + len(r.ext)) * grow + 64
        while True:
            out = DynRecords.empty(cap_n, cap_b, cap_b)
            ci, co = r._c(), out._c()
            co.cap_n, co.cap_key, co.cap_ext = cap_n, cap_b, cap_b
            t0 = time.perf_counter()
            st = fn(self.ctx, C.byref(ci), *make_args(co))
            self.last_call_ms = (time.perf_counter() - t0) * 1e3    # inside the C ABI (the last attempt): what a C / Java host pays
            if st == RFX_E_CAP:
                cap_n = max(cap_n, int(co.n)); cap_b = max(cap_b, int(co.need_key), int(co.need_ext)) + 64
                continue
            self._check(st, name)
            return out._trim(co)

    def dyn_binarize(self, rows, form=None) -> DynRecords:
        """rfx_dyn_binarize: DynamicKmerBinarizerFromReducedToSubKmer ON THE DEVICE.  rows: tuples of text fields -- (k-mer, "m|l|r")
        (form 0, FirstFour) or (sub-k-mer, "m|l|r", extension) (form 1, Iteration); joined by ',' as the hand-over files hold them."""
        rows = list(rows)
        if form is None:
            form = 1 if rows and len(rows[0]) == 3 else 0
        lines = [",".join(f) for f in rows]
        off = np.zeros(len(lines) + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in lines])
        text = "".join(lines).encode()
        cap_n, cap_b = len(lines), len(text) + 64
        out = DynRecords.empty(cap_n, cap_b, cap_b)
        co = out._c()
        co.cap_n, co.cap_key, co.cap_ext = cap_n, cap_b, cap_b
        st = self.L.rfx_dyn_binarize(self.ctx, text, _p(off), C.c_int64(len(lines)), form, C.byref(co))
        self._check(st, "rfx_dyn_binarize")
        return out._trim(co)

    def dyn_sort(self, r: DynRecords, P: int):
        """rfx_dyn_sort -> (sorted DynRecords, part_start[P+1])"""
        ps = np.empty(P + 1, np.int64)
        out = self._dyn_call(self.L.rfx_dyn_sort, "rfx_dyn_sort", r, lambda co: (P, C.byref(co), _p(ps)))
        return out, ps

    def dyn_random_reflection(self, r: DynRecords, part_start):
        part_start = np.ascontiguousarray(part_start, np.int64)
        return self._dyn_call(self.L.rfx_dyn_random_reflection, "rfx_dyn_random_reflection", r,
                              lambda co: (_p(part_start), len(part_start) - 1, C.byref(co)))

    def dyn_extend_pass(self, r: DynRecords, part_start, stage=0, start_iteration=5, start_marker=2):
        part_start = np.ascontiguousarray(part_start, np.int64)
        P = len(part_start) - 1
        ops = np.empty(P + 1, np.int64)
        out = self._dyn_call(self.L.rfx_dyn_extend_pass, "rfx_dyn_extend_pass", r,
                             lambda co: (_p(part_start), P, stage, start_iteration, start_marker, C.byref(co), _p(ops)))
        return out, ops

    def dyn_run(self, r: DynRecords, P=1, random_reflection=False, passes_first_four=0, start_iteration=1, end_iteration=0):
        """rfx_dyn_run (records resident in HBM between the operators) -> (DynRecords, [records after each pass])"""
        trace = np.zeros(256, np.int64)
        ntr = C.c_int64(0)
        out = self._dyn_call(self.L.rfx_dyn_run, "rfx_dyn_run", r,
                             lambda co: (P, int(random_reflection), passes_first_four, start_iteration, end_iteration, C.byref(co), _p(trace),
                                         C.c_int64(len(trace)), C.byref(ntr)))
        return out, [int(x) for x in trace[:ntr.value]]

    # ------------------------------------------------ f-4: contig RC de-duplication
    def dedup_contigs(self, contigs, min_contig=500):
        """rfx_dedup_contigs (P/ReflexivDSDynamicKmerDedup.java :138-339) on a list of contig strings (ids = positions) ->
        (survivors: list of strings, text, [contigs after round 1, 2, 3])"""
        off = np.zeros(len(contigs) + 1, np.int64)
        off[1:] = np.cumsum([len(c) for c in contigs])
        bases = np.frombuffer("".join(contigs).encode(), np.uint8) if off[-1] else np.zeros(1, np.uint8)
        n = len(contigs)
        ob = np.empty(int(2 * off[-1]) + 4096, np.uint8)
        oo = np.empty(n + 2, np.int64)
        m, tl = C.c_int64(0), C.c_int64(0)
        rn = (C.c_int64 * 3)()
        tcap = int(2 * off[-1]) + 64 * (n + 2) + 4096
        tb = np.empty(tcap, np.uint8)
        t0 = time.perf_counter()
        st = self.L.rfx_dedup_contigs(self.ctx, _p(bases), _p(off), C.c_int64(n), min_contig, _p(ob), C.c_int64(len(ob)), _p(oo),
                                      C.c_int64(n + 1), C.byref(m), _p(tb), C.c_int64(tcap), C.byref(tl), rn)
        self.last_call_ms = (time.perf_counter() - t0) * 1e3        # inside the C ABI: what a C / Java host pays
        self._check(st, "rfx_dedup_contigs")
        k = int(m.value)
        surv = [bytes(ob[oo[i]:oo[i + 1]]).decode() for i in range(k)]
        return surv, bytes(tb[:tl.value]).decode(), [int(x) for x in rn]

    def dedup_contig_text(self, text: str, min_contig=500):
        """rfx_dedup_contig_text: the path's contig text -> (de-duplicated text, contigs, [after round 1, 2, 3])"""
        src = text.encode()
        cap = len(src) + 4096
        out = np.empty(cap, np.uint8)
        ln, nc = C.c_int64(0), C.c_int64(0)
        rn = (C.c_int64 * 3)()
        self._check(self.L.rfx_dedup_contig_text(self.ctx, src, C.c_int64(len(src)), min_contig, _p(out), C.c_int64(cap), C.byref(ln),
                                                 C.byref(nc), rn), "rfx_dedup_contig_text")
        return bytes(out[:ln.value]).decode(), int(nc.value), [int(x) for x in rn]

    # ------------------------------------------------ several GPUs: the RCCL exchange behind the C ABI
    @staticmethod
    def comm_unique_id() -> bytes:
        """rank 0: the 128-byte RCCL id every rank passes to comm_init (hand it round with whatever the host has)"""
        buf = (C.c_uint8 * 128)()
        st = _lib.lib().rfx_comm_unique_id(buf)
        if st != RFX_OK:
            raise RfxError(st, "rfx_comm_unique_id", "RCCL is not available")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """collective: -> this rank's communicator handle (kept by the object; comm_destroy / close free it)"""
        assert len(unique_id) == 128
        h = C.c_void_p(0)
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.L.rfx_comm_init(self.ctx, buf, rank, world, C.byref(h)), "rfx_comm_init")
        self.comm = h
        self.comm_rank, self.comm_world = rank, world
        return h

    def trim(self):
        """rfx_ctx_trim: the context's grow-only workspaces back to the driver"""
        self._check(self.L.rfx_ctx_trim(self.ctx), "rfx_ctx_trim")

    def workspace_bytes(self) -> int:
        return int(self.L.rfx_ctx_workspace_bytes(self.ctx))

    def comm_destroy(self):
        if getattr(self, "comm", None):
            self.L.rfx_comm_destroy(self.comm)
            self.comm = None

    def comm_all_reduce(self, vals, op: str = "sum"):
        a = (C.c_int64 * len(vals))(*[int(v) for v in vals])
        self._check(self.L.rfx_comm_all_reduce_i64(self.comm, a, len(vals), 0 if op == "sum" else 1), "rfx_comm_all_reduce_i64")
        return [int(x) for x in a]

    def sharded_count_dev(self, d_words: int, n_reads: int, words_per_read: int, read_len: int, k: int, d_out_keys: int,
                          d_out_counts: int, cap: int, min_cov=2, max_cov=10_000_000, twin=TWIN_DS, generations=4,
                          front_clip=0, end_clip=0, d_read_len: int = 0):
        """collective (rfx_dev_sharded_count): this rank's reads -> its ascending shard; -> (m, [instances, distinct,
        survivors] over all ranks).  RfxError(RFX_E_CAP) carries the needed capacity in .need."""
        n = C.c_int64(0)
        tot = (C.c_int64 * 3)()
        st = self.L.rfx_dev_sharded_count(self.ctx, self.comm, C.c_void_p(d_words), C.c_void_p(d_read_len), C.c_int64(n_reads), words_per_read,
                                          read_len, k, front_clip, end_clip, generations, min_cov, max_cov, twin,
                                          C.c_void_p(d_out_keys), C.c_void_p(d_out_counts), C.c_int64(cap), C.byref(n), tot)
        if st == RFX_E_CAP:
            e = RfxError(st, "rfx_dev_sharded_count", f"needs room for {n.value} survivors, cap is {cap}")
            e.need = int(n.value)
            raise e
        self._check(st, "rfx_dev_sharded_count")
        return int(n.value), [int(x) for x in tot]

    def sharded_assemble_dev(self, d_keys: int, d_counts: int, n: int, prm: Params, gather_below: int = -1, text_cap: int = None):
        """collective (rfx_dev_sharded_assemble): this rank's shard of the filtered (k-mer, count) list in HBM -> (text, n_contigs,
        trace) on rank 0 ("" elsewhere), the extend stage range-sharded over the ranks while the record set has more than
        gather_below records (-1: the library's default, 0: to the end of the loop)"""
        trace = np.zeros(prm.max_iter + 8, np.int64)
        cap = (1 << 20) + 64 * max(n, 1) if text_cap is None else int(text_cap)
        self.text_retries = 0
        while True:
            buf = np.empty(max(cap, 1), np.uint8)
            ln, nc, ntr = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            st = self.L.rfx_dev_sharded_assemble(self.ctx, self.comm, C.c_void_p(d_keys), C.c_void_p(d_counts), C.c_int64(n), C.byref(prm),
                                                 C.c_int64(gather_below), _p(buf), C.c_int64(cap), C.byref(ln), C.byref(nc), _p(trace),
                                                 C.c_int64(len(trace)), C.byref(ntr))
            if st == RFX_E_CAP and ln.value > cap:
                cap = int(ln.value)
                self.text_retries += 1
                continue
            self._check(st, "rfx_dev_sharded_assemble")
            return bytes(buf[:ln.value]).decode(), int(nc.value), [int(x) for x in trace[:ntr.value]]

    def sharded_assemble_reads(self, bases, read_off, prm: Params, generations: int = 4, text_cap: int = None, gather_below: int = -1):
        """collective (rfx_sharded_assemble_reads): this rank's ASCII reads -> (text, n_contigs, trace, totals); text on rank 0.
        A text buffer that is too short on rank 0 is RFX_E_CAP on EVERY rank (with the length rank 0 needs), so the retry below
        re-enters the collective on all ranks together; text_cap forces a first size (tests)."""
        bases = np.ascontiguousarray(bases, np.uint8)
        read_off = np.ascontiguousarray(read_off, np.int64)
        n_reads = len(read_off) - 1
        trace = np.zeros(prm.max_iter + 8, np.int64)
        tot = (C.c_int64 * 3)()
        cap = 3 * len(bases) + (1 << 20) if text_cap is None else int(text_cap)
        self.text_retries = 0
        while True:
            buf = np.empty(max(cap, 1), np.uint8)
            ln, nc, ntr = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            st = self.L.rfx_sharded_assemble_reads(self.ctx, self.comm, _p(bases), _p(read_off), C.c_int64(n_reads), C.byref(prm),
                                                   generations, C.c_int64(gather_below), _p(buf), C.c_int64(cap), C.byref(ln), C.byref(nc), _p(trace),
                                                   C.c_int64(len(trace)), C.byref(ntr), tot)
            if st == RFX_E_CAP and ln.value > cap:
                cap = int(ln.value)
                self.text_retries += 1
                continue
            self._check(st, "rfx_sharded_assemble_reads")
            return (bytes(buf[:ln.value]).decode(), int(nc.value), [int(x) for x in trace[:ntr.value]], [int(x) for x in tot])

    def comm_bytes_bucketed(self) -> int:
        return int(self.L.rfx_comm_last_bytes_bucketed(self.comm))

    def gather_shards_dev(self, d_keys: int, d_counts: int, n: int, key_words: int, count_bytes: int, root: int,
                          d_out_keys: int, d_out_counts: int, cap: int) -> int:
        m = C.c_int64(0)
        st = self.L.rfx_dev_gather_shards(self.ctx, self.comm, C.c_void_p(d_keys), C.c_void_p(d_counts), C.c_int64(n), key_words,
                                          count_bytes, root, C.c_void_p(d_out_keys), C.c_void_p(d_out_counts), C.c_int64(cap),
                                          C.byref(m))
        if st == RFX_E_CAP:
            e = RfxError(st, "rfx_dev_gather_shards", f"needs room for {m.value} entries, cap is {cap}")
            e.need = int(m.value)
            raise e
        self._check(st, "rfx_dev_gather_shards")
        return int(m.value)

    def count_timing(self):
        """Per-kernel-family HIP-event timing of the last count call: {name: (ms, launches)}; the "stat_*" entries carry
        the leaf tables' statistics in `launches` (leaves, table passes, passes abandoned on overflow) and the last level's
        (records of children that outgrew their regions and were moved, sweeps given up for the exact form)."""
        out = {}
        for name in ("hist1", "part1", "hist2", "part2", "hist3", "part3", "leaf", "sort", "extract_w", "count_w",
                     "pair_hist", "pair_part", "stat_leaves", "stat_passes", "stat_overflows", "stat_l2_spilled", "stat_l2_void"):
            ms, ln = C.c_float(0), C.c_int64(0)
            if self.L.rfx_last_count_timing(self.ctx, name.encode(), C.byref(ms), C.byref(ln)) == RFX_OK:
                out[name] = (float(ms.value), int(ln.value))
        return out
