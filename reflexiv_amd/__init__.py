"""reflexiv_amd -- MI355X-native k-mer counting + reflexible extend-and-merge (Reflexiv hot path).

The package is a thin host layer over libreflexiv_hip.so (HIP kernels for gfx950 behind the C ABI
of include/reflexiv_hip.h).  There is no CPU implementation in here: the CPU oracle under oracle/
is test infrastructure and is never imported by this package.
"""
from ._lib import RfxError, Params, TWIN_DS, TWIN_RDD, build, lib  # noqa: F401
from .api import Reflexiv, Records, default_params, as_records  # noqa: F401

__all__ = ["Reflexiv", "Records", "Params", "RfxError", "default_params", "as_records", "TWIN_DS", "TWIN_RDD",
           "build", "lib"]
