"""ctypes binding of libreflexiv_hip.so (include/reflexiv_hip.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 GPU is
present when a context is created, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libreflexiv_hip.so")
_LIB = None

RFX_OK, RFX_E_ARG, RFX_E_CAP, RFX_E_HIP, RFX_E_NOGPU, RFX_E_STATE, RFX_E_LIMIT, RFX_E_HOST = 0, -1, -2, -3, -4, -5, -6, -7
TWIN_DS, TWIN_RDD = 0, 1
_STATUS = {0: "RFX_OK", -1: "RFX_E_ARG", -2: "RFX_E_CAP", -3: "RFX_E_HIP", -4: "RFX_E_NOGPU",
           -5: "RFX_E_STATE", -6: "RFX_E_LIMIT", -7: "RFX_E_HOST"}


class RfxError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: {_STATUS.get(status, status)}{(' -- ' + detail) if detail else ''}")


class Params(C.Structure):
    """rfx_params (U/DefaultParam.java:74-120)."""
    _fields_ = [(n, C.c_int32) for n in (
        "k", "min_cov", "max_cov", "min_error_cov", "min_contig", "min_iter", "max_iter",
        "front_clip", "end_clip", "partitions", "twin", "coalesce", "extras")]


class CDynRecords(C.Structure):
    """rfx_dyn_records."""
    _fields_ = [("n", C.c_int64), ("key", C.c_void_p), ("key_off", C.c_void_p), ("ext", C.c_void_p), ("ext_off", C.c_void_p),
                ("marker", C.c_void_p), ("left", C.c_void_p), ("right", C.c_void_p), ("cap_n", C.c_int64), ("cap_key", C.c_int64),
                ("cap_ext", C.c_int64), ("need_key", C.c_int64), ("need_ext", C.c_int64)]


class CRecords(C.Structure):
    """rfx_records."""
    _fields_ = [("n", C.c_int64), ("key", C.c_void_p), ("marker", C.c_void_p), ("ext_off", C.c_void_p),
                ("ext", C.c_void_p), ("left", C.c_void_p), ("right", C.c_void_p),
                ("cap_n", C.c_int64), ("cap_words", C.c_int64), ("need_n", C.c_int64), ("need_words", C.c_int64),
                ("key_words", C.c_int32), ("reserved_", C.c_int32)]


# every symbol include/reflexiv_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "rfx_version", "rfx_default_params", "rfx_ctx_create", "rfx_ctx_destroy", "rfx_ctx_sync",
    "rfx_ctx_set_stream", "rfx_ctx_stream", "rfx_last_error", "rfx_ctx_trim", "rfx_ctx_workspace_bytes",
    "rfx_extract_canon", "rfx_count_filter", "rfx_rc_expand_subkmer", "rfx_sort_records",
    "rfx_fork_filter_forward", "rfx_reflect_from_forward", "rfx_fork_filter_reflected",
    "rfx_random_reflection", "rfx_extend_pass", "rfx_contigs_text",
    "rfx_dev_encode_reads", "rfx_kmers_per_read", "rfx_count_workspace_bytes", "rfx_dev_count_reads",
    "rfx_dev_count_kmers", "rfx_dev_bucket_by_owner", "rfx_dev_bucket_records_by_owner",
    "rfx_dev_count_records", "rfx_dev_assemble", "rfx_dev_synth_genome",
    "rfx_dev_synth_reads", "rfx_dev_sort_pairs", "rfx_last_count_timing",
    "rfx_extract_canon_w", "rfx_count_filter_w", "rfx_kmers_per_read_w", "rfx_dev_count_reads_w",
    "rfx_dev_count_reads_ragged", "rfx_assemble_reads", "rfx_dev_bucket_wide_by_owner", "rfx_dev_count_wide_elems",
    "rfx_dev_combine_reads", "rfx_dev_bucket_pairs_by_owner", "rfx_dev_merge_pairs",
    "rfx_dev_bucket_wide_records_by_owner", "rfx_dev_count_wide_records",
    "rfx_extras_operator", "rfx_assemble_counts_w", "rfx_dev_rc_expand_subkmer", "rfx_dev_sort_records", "rfx_dev_fork_filter",
    "rfx_dev_reflect_from_forward", "rfx_dev_random_reflection", "rfx_dev_extend_pass", "rfx_dev_lower_bound", "rfx_extend_pass_w", "rfx_dev_counter_to_asm", "rfx_dev_assemble_w", "rfx_dev_order_kmers_w",
    "rfx_comm_unique_id", "rfx_comm_init", "rfx_comm_destroy", "rfx_comm_rank", "rfx_comm_world", "rfx_comm_last_bytes_bucketed",
    "rfx_comm_all_reduce_i64", "rfx_dev_sharded_count", "rfx_dev_gather_shards", "rfx_sharded_assemble_reads", "rfx_dev_sharded_assemble",
    "rfx_dedup_contigs", "rfx_dedup_contig_text",
    "rfx_dyn_binarize", "rfx_dyn_sort", "rfx_dyn_random_reflection", "rfx_dyn_extend_pass", "rfx_dyn_run",
    "rfx_dyn_blocks_to_bases", "rfx_dyn_bases_to_blocks", "rfx_dyn_attribute", "rfx_dyn_attribute_unpack",
]


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src)
                 if f.endswith((".hip", ".h")) or f == "Makefile")
    newest = max(newest, os.path.getmtime(os.path.join(_HERE, "..", "include", "reflexiv_hip.h")))
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-s", "-j4", "-C", src])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RfxError(RFX_E_NOGPU, "reflexiv_amd",
                           f"{LIB_PATH} is missing -- build it with reflexiv_amd._lib.build() "
                           "(there is no CPU fallback)")
        # One HIP runtime per process: torch ships its own libamdhip64 (soname libamdhip64.so.7).
        # Loaded first, it also satisfies this library's NEEDED entry, so device pointers and
        # streams are shared with torch; loaded second, the process would hold two runtimes.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.rfx_version.restype = C.c_int
        L.rfx_ctx_stream.restype = C.c_void_p
        L.rfx_last_error.restype = C.c_char_p
        L.rfx_kmers_per_read.restype = C.c_int64
        L.rfx_count_workspace_bytes.restype = C.c_int64
        L.rfx_count_workspace_bytes.argtypes = [C.c_int64]
        for name in SYMBOLS:
            fn = getattr(L, name)
            if name == "rfx_kmers_per_read_w":
                fn.restype = C.c_int64
                continue
            if name == "rfx_dyn_attribute":
                fn.restype = C.c_int64
                fn.argtypes = [C.c_int, C.c_int, C.c_int]
                continue
            if name == "rfx_dyn_attribute_unpack":
                fn.restype = None
                continue
            if name in ("rfx_comm_last_bytes_bucketed", "rfx_ctx_workspace_bytes"):
                fn.restype = C.c_int64
                fn.argtypes = [C.c_void_p]
                continue
            if name == "rfx_comm_destroy":
                fn.restype = None
                fn.argtypes = [C.c_void_p]
                continue
            if name not in ("rfx_ctx_stream", "rfx_last_error", "rfx_kmers_per_read",
                            "rfx_count_workspace_bytes", "rfx_ctx_destroy", "rfx_default_params"):
                fn.restype = C.c_int
        _LIB = L
    return _LIB
